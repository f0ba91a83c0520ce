"""Drop-in for reference src/loss_functions/VisibilityPriorLoss01.py (the paper's visibility-prior term)."""
import torch

from loss_functions.FusedLossesHip01 import fused_loss_values


class VisibilityPriorLossHip:
    FUSED_SLOTS = (4, 5)                      # (coarse, fine) entries of the fused loss vector this class reports

    def __init__(self, configs: dict, loss_configs: dict):
        self.configs, self.loss_configs = configs, loss_configs
        self.fine_mlp_needed = 'fine_mlp' in configs['model']

    def compute_loss(self, input_dict: dict, output_dict: dict, return_loss_maps: bool = False):
        if 'raw_visibility2_coarse' not in output_dict or (self.fine_mlp_needed and 'raw_visibility2_fine' not in output_dict):
            return None                               # validation frames rendered without secondary views
        v = fused_loss_values(self.configs, input_dict, output_dict)
        loss_dict = {'loss_value': v[4] + v[5] if self.fine_mlp_needed else v[4]}
        if return_loss_maps:
            m = input_dict['indices_mask_nerf']
            pw = input_dict.get('visibility_prior_masks', input_dict.get('visibility_prior_weights'))
            maps = {}
            for lv in ('coarse', 'fine') if self.fine_mlp_needed else ('coarse',):
                v2 = output_dict[f'visibility2_{lv}'][m]
                w = pw[m] if pw is not None else torch.ones_like(v2)
                maps[f'VisibilityPriorLossHip01_{lv}'] = torch.sum(w * (1 - v2), dim=1)
            loss_dict['loss_maps'] = maps
        return loss_dict
