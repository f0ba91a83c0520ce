"""A compact trainer around the HIP path with the reference trainer's sequence (reference src/Trainer01.py):

    train()           :265-311  per iteration: learning-rate decay -> train_one_iter -> [validation] -> [checkpoint]
    train_one_iter()  :61-107   next batch -> zero_grad -> per sub-batch: forward, compute_losses, TotalLoss.backward -> step
    run_validation()  :109-263  eval-mode render of whole frames (here: camera -> uint8 image on the GPU, predict_frame)
    save / load_model :352-381  the reference's checkpoint files (CheckpointHip01)
    learning rate     src/lr_decayers/NeRFLearningRateDecayer01.py:14-24   lr_init * 0.1 ** (iter / (lr_decay * 1000))

The reference's Trainer is a Python harness around tensorboard / skimage / its dataset loaders and stays what it is inside the
reference tree (the model and loss classes drop into it by name).  This file is the same loop for use WITHOUT
that tree: batches come from RayGeneratorHip + BatchIndexScheduler (on-device ray generation, the reference's index schedule),
several GPUs are one process each with vipnerf_hip.dist (row-class-aware shards, one all-reduce of the flat gradient bucket),
and nothing of the per-iteration work runs on the host.  No CPU fallback anywhere.
"""
import os
import sys
from pathlib import Path

import numpy
import torch

try:
    import vipnerf_hip  # noqa: F401
except ImportError:
    for cand in (os.environ.get('VIPNERF_HIP_ROOT'), str(Path(__file__).resolve().parents[1])):
        if cand and cand not in sys.path:
            sys.path.insert(0, cand)
from vipnerf_hip import dist as vdist
from vipnerf_hip.optim import FlatAdam

import CheckpointHip01 as ckpt
from data_preprocessors.RayGeneratorHip01 import BatchIndexScheduler, RayGeneratorHip, predict_frame
from loss_functions.LossComputerHip01 import LossComputerHip
from models.ModelFactory import get_model


class TrainerHip:
    def __init__(self, configs: dict, ray_generator: RayGeneratorHip, scheduler: BatchIndexScheduler, output_dirpath=None,
                 rank: int = 0, world: int = 1):
        """configs: the reference's config dict (`model`, `losses`, `data_loader.ndc`, `optimizer.{lr_initial, lr_decay, beta1,
        beta2}`, `num_iterations`, optional `sub_batch_size`, `validation_interval`, `model_save_interval`)."""
        self.configs, self.gen, self.scheduler = configs, ray_generator, scheduler
        self.rank, self.world = rank, world
        self.device = ray_generator.device
        self.output_dirpath = Path(output_dirpath) if output_dirpath is not None else None
        self.model = get_model(configs, None).to(self.device)
        vdist.broadcast_parameters(self.model)
        self.loss_computer = LossComputerHip(configs)
        oc = configs.get('optimizer', {})
        self.lr_init, self.lr_decay_steps = float(oc.get('lr_initial', 5e-4)), float(oc.get('lr_decay', 250)) * 1000
        self.optimizer = FlatAdam(self.model.parameters(), lr=self.lr_init, betas=(oc.get('beta1', 0.9), oc.get('beta2', 0.999)))
        self.bucket = vdist.FlatGradBucket(self.model.parameters())
        # one library call per iteration (vipnerf_train_step) where an iteration is one sub-batch of the fused ray losses -- what the
        # reference's shipped configs are (1024, or 2048 + 2048 rays per iteration: the batch sizes at which five Python -> ctypes calls per
        # iteration would bound the step); `one_call_step: False` keeps the module-contract sequence
        self.stepper = None
        if configs.get('one_call_step', True):
            from vipnerf_hip._lib import VipNerfHipError
            from vipnerf_hip.step import FusedTrainStep
            try:
                self.stepper = FusedTrainStep(self.model, configs, self.optimizer, bucket_reduce=vdist.all_reduce_mean_flat if world > 1 else None)
            except VipNerfHipError:              # a loss outside the fused four: the module contract serves it
                self.stepper = None

    def learning_rate(self, iter_num: int) -> float:
        return self.lr_init * (0.1 ** (iter_num / self.lr_decay_steps))

    def train_one_iter(self, iter_num: int) -> dict:
        """-> {loss name: 0-dim device tensor, summed over the sub-batches}; nothing here waits for the GPU."""
        batch = self.gen.get_next_batch(iter_num, scheduler=self.scheduler)
        if self.world > 1:
            # a short last batch of an epoch / an odd number of sparse-depth pixels: each row class is trimmed to a multiple of the
            # ranks (equal per-class counts keep the mean of the rank means exact) instead of stopping the run
            batch = vdist.shard_batch(batch, self.rank, self.world, uneven='trim')
        self.bucket.release()                                # = optimizer.zero_grad(set_to_none=True)
        n = batch['rays_o'].shape[0]
        if n == 0:
            # a row class with fewer rows than ranks trims to nothing (the very short last batch of an epoch): every rank sees the same
            # host-side counts, so every rank skips this iteration -- no backward, no collective, no optimizer step, identically everywhere
            return {}
        sub = max(1, int(self.configs.get('sub_batch_size', n)) or n)
        if self.stepper is not None and sub >= n:
            from vipnerf_hip.step import named_losses
            return named_losses(self.stepper(batch))
        logged = {}
        for s in range(0, n, sub):
            sb = {k: (v[s:s + sub] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}
            sb['common_data'] = dict(batch['common_data'])
            if 'rng_ray_ids' not in sb:
                sb['rng_ray_base'] = s                       # sub-batches draw the rows' own numbers of the batch's streams
            out = self.model(sb)
            losses = self.loss_computer.compute_losses(sb, out)
            losses['TotalLoss'].backward()
            for k, v in losses.items():                      # kept on the device: a float() here would stall the host every iteration
                v = v['loss_value'] if isinstance(v, dict) else v
                v = v.detach() if isinstance(v, torch.Tensor) else torch.as_tensor(float(v), device=self.device)
                logged[k] = logged[k] + v if k in logged else v
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        return logged

    def run_validation(self, frames=None) -> dict:
        """Eval-mode render of the generator's cameras; PSNR against their images when the generator holds them."""
        self.model.eval()
        res = {}
        for f in (range(self.gen.n) if frames is None else frames):
            r = predict_frame(self.model, self.gen, frame=f)
            if self.gen.images is not None:
                ref = self.gen.images[f]
                mse = torch.mean((r['image'].float() / 255 - ref) ** 2)
                r['psnr'] = float(-10 * torch.log10(mse.clamp_min(1e-12)))
            res[f] = r
        self.model.train()
        return res

    def save_model(self, iter_num: int):
        if self.rank == 0 and self.output_dirpath is not None:
            return ckpt.save_model(self.model, self.optimizer, iter_num, self.output_dirpath)

    def load_model(self) -> int:
        latest = None if self.output_dirpath is None else self.output_dirpath / 'saved_models' / 'Model_Latest.tar'
        if latest is None or not latest.exists():
            return 0
        return ckpt.load_model(self.model, latest, self.optimizer, map_location=self.device)

    def train(self, log_every: int = 0) -> list:
        total = int(self.configs['num_iterations'])
        val_int = int(self.configs.get('validation_interval', 0))
        save_int = int(self.configs.get('model_save_interval', 0))
        start = self.load_model()
        history, pending = [], []

        def flush():
            # one transfer per block of iterations: the loop never waits for the GPU per step, and a run of 250 k iterations neither keeps
            # millions of 0-dim device tensors alive nor loses its whole log if it dies
            if not pending:
                return
            rows = [h for h in pending if h[0]]                  # (loss-less rows: a skipped iteration on a validation boundary)
            values = {}
            if rows:
                keys = list(rows[0][0].keys())
                table = torch.stack([torch.stack([h[0][k].float() for k in keys]) for h in rows]).cpu().numpy()
                values = {id(h[1]): dict(zip(keys, map(float, row))) for row, h in zip(table, rows)}
            history.extend(dict(values.get(id(h[1]), {}), **h[1]) for h in pending)
            pending.clear()

        flush_every = int(self.configs.get('log_flush_interval', 256))
        self.model.train()
        for iter_num in range(start, total):
            lr = self.learning_rate(iter_num)
            for g in self.optimizer.param_groups:
                g['lr'] = lr
            losses = self.train_one_iter(iter_num)
            # every row names its iteration: after a skipped iteration the row position is no longer the iteration number
            entry = {'iter': iter_num, 'lr': lr}
            validate = bool(val_int) and (iter_num + 1) % val_int == 0
            if losses or validate:                           # a skipped iteration (empty trimmed shard) logs no losses; one that falls on a
                pending.append((losses, entry))              # validation boundary still gets its (loss-less) row, so that the psnr is not lost
            if log_every and losses and self.rank == 0 and (iter_num + 1) % log_every == 0:
                print(f"iter {iter_num + 1}: " + ' '.join(f'{k} {float(v):.5f}' for k, v in losses.items()) + f' lr {lr:.3e}', flush=True)
            if validate:
                entry['validation_psnr'] = float(numpy.mean([v.get('psnr', float('nan')) for v in self.run_validation().values()]))
            if save_int and (iter_num + 1) % save_int == 0:
                self.save_model(iter_num + 1)
            if len(pending) >= flush_every or (val_int and (iter_num + 1) % val_int == 0) or (save_int and (iter_num + 1) % save_int == 0):
                flush()
        flush()
        return history
