"""Drop-in for reference src/loss_functions/MSE01.py (class found by LossComputer's name[:-2] rule)."""
import torch

from loss_functions.FusedLossesHip01 import fused_loss_values


class MSEHip:
    FUSED_SLOTS = (0, 1)                      # (coarse, fine) entries of the fused loss vector this class reports

    def __init__(self, configs: dict, loss_configs: dict):
        self.configs, self.loss_configs = configs, loss_configs
        self.fine_mlp_needed = 'fine_mlp' in configs['model']

    def compute_loss(self, input_dict: dict, output_dict: dict, return_loss_maps: bool = False):
        v = fused_loss_values(self.configs, input_dict, output_dict)
        loss_dict = {'loss_value': v[0] + v[1] if self.fine_mlp_needed else v[0]}
        if return_loss_maps:                      # validation-only convenience, plain tensor ops
            m = input_dict['indices_mask_nerf']
            maps = {}
            for lv in ('coarse', 'fine') if self.fine_mlp_needed else ('coarse',):
                e = output_dict[f'rgb_{lv}'][m] - input_dict['target_rgb'][m]
                maps[f'MSEHip01_{lv}'] = torch.mean(torch.square(e), dim=1)
            loss_dict['loss_maps'] = maps
        return loss_dict
