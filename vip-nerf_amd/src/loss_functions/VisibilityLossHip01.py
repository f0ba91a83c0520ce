"""Drop-in for reference src/loss_functions/VisibilityLoss01.py."""
import torch

from loss_functions.FusedLossesHip01 import fused_loss_values


class VisibilityLossHip:
    FUSED_SLOTS = (2, 3)                      # (coarse, fine) entries of the fused loss vector this class reports

    def __init__(self, configs: dict, loss_configs: dict):
        self.configs, self.loss_configs = configs, loss_configs
        self.fine_mlp_needed = 'fine_mlp' in configs['model']

    def compute_loss(self, input_dict: dict, output_dict: dict, return_loss_maps: bool = False):
        if 'raw_visibility_coarse' not in output_dict:            # mlp predict_visibility = False: what VisibilityLoss01.py raises there
            raise KeyError('raw_visibility_coarse')
        v = fused_loss_values(self.configs, input_dict, output_dict)
        loss_dict = {'loss_value': v[2] + v[3] if self.fine_mlp_needed else v[2]}
        if return_loss_maps:
            maps = {}
            for lv in ('coarse', 'fine') if self.fine_mlp_needed else ('coarse',):
                e = (output_dict[f'raw_visibility_{lv}'][..., 0] - output_dict[f'visibility_{lv}']).detach()
                maps[f'VisibilityLossHip01_{lv}'] = 2 * torch.mean(torch.abs(e), dim=1)
            loss_dict['loss_maps'] = maps
        return loss_dict
