"""Device glue with the reference's contract (reference src/utils/CommonUtils01.py:15-42).

get_device(device): `device` is None / '' / an index / a list of indices (the reference's configs['device']); the first
index is chosen.  Like the reference it answers torch.device('cpu') when no GPU is usable -- the HIP model itself then
refuses to run (there is no CPU fallback in this tree), with a message that says so.
move_to_device(data, device): recursive `.to(device, non_blocking=True)` over tensors, lists and dicts; everything else is
passed through unchanged.
"""
from typing import Union

import torch


def get_device(device):
    if (device is None) or (device == '') or (not torch.cuda.is_available()):
        return torch.device('cpu')
    first = device[0] if isinstance(device, (list, tuple)) else device
    return torch.device(f'cuda:{first}')


def move_to_device(tensor_data: Union[torch.Tensor, list, dict], device):
    if isinstance(tensor_data, torch.Tensor):
        return tensor_data.to(device, non_blocking=True)
    if isinstance(tensor_data, list):
        return [move_to_device(t, device) for t in tensor_data]
    if isinstance(tensor_data, dict):
        return {k: move_to_device(v, device) for k, v in tensor_data.items()}
    return tensor_data
