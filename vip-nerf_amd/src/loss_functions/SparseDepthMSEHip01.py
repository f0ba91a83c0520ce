"""Drop-in for reference src/loss_functions/SparseDepthMSE01.py."""
from loss_functions.FusedLossesHip01 import fused_loss_values


class SparseDepthMSEHip:
    FUSED_SLOTS = (6, 7)                      # (coarse, fine) entries of the fused loss vector this class reports

    def __init__(self, configs: dict, loss_configs: dict):
        self.configs, self.loss_configs = configs, loss_configs

    def compute_loss(self, input_dict: dict, output_dict: dict, return_loss_maps: bool = False):
        v = fused_loss_values(self.configs, input_dict, output_dict)
        loss_dict = {'loss_value': v[6] if 'indices_mask_sparse_depth' in input_dict else v[6].detach() * 0}
        if return_loss_maps:
            loss_dict['loss_maps'] = {}
        return loss_dict
