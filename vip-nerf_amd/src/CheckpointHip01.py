"""Checkpoint compatibility with the reference trainer.

The reference saves `{'iteration_num', 'model_state_dict', 'optimizer_state_dict'}` with torch.save to
`saved_models/Model_Iter{N:06}.tar` plus a `Model_Latest.tar` symlink (reference src/Trainer01.py:352-366) and
loads it back at the start of training / testing (:368-381, src/Tester01.py:45-49).  Because the model is wrapped in
torch.nn.DataParallel (:517) every key carries a `module.` prefix.  VipNeRFHip has the reference's parameter names
and shapes, so the authors' released weights load directly; these helpers only deal with the prefix and the file
layout.
"""
import os
from pathlib import Path

import torch

PREFIX = 'module.'


def strip_prefix(state_dict: dict) -> dict:
    return {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in state_dict.items()}


def add_prefix(state_dict: dict) -> dict:
    return {(k if k.startswith(PREFIX) else PREFIX + k): v for k, v in state_dict.items()}


def load_model(model: torch.nn.Module, path, optimizer=None, map_location='cpu') -> int:
    """Loads a reference (or own) checkpoint into `model` (wrapped in DataParallel or not).  Returns iteration_num."""
    ckpt = torch.load(str(path), map_location=map_location, weights_only=False)
    sd = ckpt['model_state_dict']
    target = model.module if isinstance(model, torch.nn.DataParallel) else model
    target.load_state_dict(strip_prefix(sd), strict=True)
    if optimizer is not None and 'optimizer_state_dict' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return int(ckpt.get('iteration_num', 0))


def save_model(model: torch.nn.Module, optimizer, iteration_num: int, output_dirpath, label: str = None) -> Path:
    """Writes the reference's checkpoint format (keys prefixed with `module.` like the DataParallel-wrapped
    reference model) and refreshes Model_Latest.tar."""
    out = Path(output_dirpath) / 'saved_models'
    out.mkdir(parents=True, exist_ok=True)
    if label is None:
        label = f'Iter{iteration_num:06}'
    target = model.module if isinstance(model, torch.nn.DataParallel) else model
    path = out / f'Model_{label}.tar'
    torch.save({'iteration_num': iteration_num, 'model_state_dict': add_prefix(target.state_dict()),
                'optimizer_state_dict': optimizer.state_dict() if optimizer is not None else {}}, str(path))
    latest = out / 'Model_Latest.tar'
    if latest.is_symlink() or latest.exists():
        latest.unlink()
    os.symlink(path.name, str(latest))
    return path
