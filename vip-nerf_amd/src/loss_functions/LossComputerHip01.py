"""Loss registry with the reference's contract (reference src/loss_functions/LossComputer01.py:12-69): losses are
looked up by module name, class = name[:-2]; TotalLoss = sum weight(iter) * loss_value; `iter_weights` picks the
last threshold <= iter_num.  Inside the reference tree the reference's own LossComputer01 does this job and finds
the *Hip01 classes by the same rule; this file makes the tree usable standalone."""
import importlib
import inspect

import torch


class LossComputerHip:
    def __init__(self, configs: dict):
        self.losses = {}
        for loss_configs in configs['losses']:
            name = loss_configs['name']
            self.losses[name] = self.get_loss_object(name, configs, loss_configs)

    @staticmethod
    def get_loss_object(loss_name, configs, loss_configs):
        module = importlib.import_module(f'loss_functions.{loss_name}')
        for name, cls in inspect.getmembers(module, inspect.isclass):
            if name == loss_name[:-2]:
                return cls(configs, loss_configs)
        raise RuntimeError(f'Unknown Loss Function: {loss_name}')

    def compute_losses(self, input_dict, output_dict, return_loss_maps: bool = False):
        if 'common_data' in input_dict:
            for key in input_dict['common_data']:
                v = input_dict['common_data'][key]
                if isinstance(v, torch.Tensor) and v.dim() > 0 and key == 'poses' and v.dim() == 4:
                    input_dict['common_data'][key] = v[0]
        loss_values, total = {}, 0
        iter_num = input_dict['iter_num']
        for name, obj in self.losses.items():
            weight = self.get_loss_weight(obj, iter_num)
            ld = obj.compute_loss(input_dict, output_dict, return_loss_maps=return_loss_maps)
            if ld is not None:
                loss_values[name] = ld
                total = total + weight * ld['loss_value']
        loss_values['TotalLoss'] = total
        return loss_values

    @staticmethod
    def get_loss_weight(loss_obj, iter_num):
        lc = loss_obj.loss_configs
        if 'weight' in lc:
            return lc['weight']
        if 'iter_weights' in lc:
            for k in sorted((int(k) for k in lc['iter_weights']), reverse=True):
                if iter_num >= k:
                    return lc['iter_weights'][str(k)]
        raise RuntimeError(f'loss_weight is None for {type(loss_obj).__name__} at iter {iter_num}')
