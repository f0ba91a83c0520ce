"""Ray-sharded data parallelism: one process per GPU, identical weights, each rank renders its own shard of the
ray batch, and ONE all-reduce (RCCL over xGMI; `nccl` backend on ROCm) of a single flat fp32 gradient bucket per
step replaces the reference's torch.nn.DataParallel replicate/scatter/gather (reference src/Trainer01.py:517,
SURVEY.md §2.1, §8e).  No activation ever crosses GPUs.

Every loss of the path is a mean over its own row subset, so with equal per-rank row counts the mean of the rank
means is the global mean and averaging the gradients is exact.
"""
import os
from typing import Iterable, List

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC for RCCL across processes (no effect once the runtime is up)

import torch
import torch.distributed as dist


# A single process normally skips the process group and every collective.  FORCE (init_from_env(force=True), or VIPNERF_FORCE_DIST=1 in
# the environment) makes a WORLD_SIZE=1 run take the multi-rank code path anyway -- backend initialisation, parameter broadcast, the
# all-reduce of the adopted flat gradient buffer, barriers -- so that RCCL's first contact with this code does not have to wait for a
# multi-GPU node (tests/test_hip_dist.py::test_rccl_single_rank_*, bench.py --force-dist).
FORCE = False


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def init_from_env(backend: str = None, force: bool = None):
    """torchrun-style initialisation.  Returns (rank, world_size, local_rank)."""
    global FORCE
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if force is None:
        force = os.environ.get('VIPNERF_FORCE_DIST', '0') not in ('', '0')
    FORCE = bool(force)
    if (world > 1 or FORCE) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('VIPNERF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradBucket:
    """Makes every parameter's .grad a view into one contiguous fp32 buffer so that the whole model's gradient
    (1,191,946 floats = 4.77 MB for coarse + fine) is reduced with a single collective."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = []
        o = 0
        for p in self.params:
            v = self.flat[o:o + p.numel()].view_as(p)
            self.views.append(v)
            o += p.numel()
        self.attach()

    def attach(self):
        """(re)point .grad at the bucket; call after optimizer.zero_grad(set_to_none=True)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        self.flat.zero_()
        self.attach()

    def release(self):
        """Alternative to zero(): drop the gradients (like optimizer.zero_grad(set_to_none=True)).  The HIP render's
        backward returns all parameter gradients as consecutive views of one buffer, which autograd then adopts
        as .grad without any fill / add / copy kernel; all_reduce_mean() reduces that buffer in place."""
        for p in self.params:
            p.grad = None

    def adopted(self):
        """The flat tensor the current .grad tensors are consecutive views of (parameter order), or None."""
        g0 = self.params[0].grad
        if g0 is None:
            return None
        st, o = g0.untyped_storage(), g0.storage_offset()
        base = st.data_ptr()
        for p in self.params:
            g = p.grad
            if (g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.storage_offset() != o
                    or g.untyped_storage().data_ptr() != base):
                return None
            o += g.numel()
        n = o - g0.storage_offset()
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, g0.storage_offset(), (n,))

    def all_reduce_mean(self):
        """mean over ranks with ONE collective.  No-op in a single process (unless FORCE)."""
        if not _active():
            return
        flat = self.flat
        if self.params[0].grad is None or self.params[0].grad.data_ptr() != self.views[0].data_ptr():
            flat = self.adopted()
            if flat is None:                      # gradients scattered over separate tensors: gather into the bucket
                torch._foreach_copy_(self.views, [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params])
                self.attach()
                flat = self.flat
        all_reduce_mean_flat(flat)


# Attribution of the collective (bench.py `allreduce_ms_per_step`): with timing on, every all_reduce_mean_flat is bracketed by a pair of
# HIP events on the stream it is issued from (torch's process group makes its own stream wait for that stream and that stream wait for the
# collective, so the pair spans queueing + the collective + the 1 / R scaling as the step sees them).  Off by default: no events.
_TIMING = None


def timing_enable(on: bool = True):
    global _TIMING
    _TIMING = [] if on else None


def timing_read():
    """-> (collectives recorded since the last read, their total milliseconds); synchronises the recorded events."""
    global _TIMING
    if _TIMING is None:
        return 0, 0.0
    n, ms = len(_TIMING), 0.0
    for a, b in _TIMING:
        if isinstance(a, float):
            ms += (b - a) * 1e3
        else:
            b.synchronize()
            ms += a.elapsed_time(b)
    _TIMING = []
    return n, ms


def all_reduce_mean_flat(flat: torch.Tensor):
    """Mean over ranks of one flat gradient buffer in place (the FusedTrainStep hook, FlatGradBucket); no-op in a single process.  RCCL
    (`nccl`) averages inside the collective (ReduceOp.AVG: no separate scaling launch); gloo has no AVG: sum, then x 1 / world."""
    if not _active():
        return
    ev = None
    if _TIMING is not None:
        if flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(flat.device))
        else:
            import time
            ev = [time.perf_counter(), 0.0]
    if dist.get_backend() == 'nccl':
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / dist.get_world_size())
    if ev is not None:
        if flat.is_cuda:
            ev[1].record(torch.cuda.current_stream(flat.device))
        else:
            import time
            ev[1] = time.perf_counter()
        _TIMING.append(tuple(ev))


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    if _active():
        for p in module.parameters():
            dist.broadcast(p.data, src=src)


def barrier():
    if _active():
        dist.barrier()


def shard_rows(n_rows: int, rank: int, world: int, uneven: str = 'raise'):
    """Equal contiguous shards of ONE row class (every loss is a mean over its own rows: with equal per-rank counts the mean of
    the rank means is the global mean).  A count that does not divide: uneven='raise' (default) refuses; 'trim' drops the last
    n_rows % world rows -- what a trainer wants for the short last batch of an epoch.  Batches that mix row classes go through
    shard_batch."""
    if n_rows % world and uneven != 'trim':
        raise ValueError(f'{n_rows} rows do not split evenly over {world} ranks: every loss is a mean over its own rows, so '
                         f'unequal shards would make the mean of the rank means differ from the global mean')
    per = n_rows // world
    return slice(rank * per, (rank + 1) * per)


def shard_row_ids(batch: dict, rank: int, world: int, uneven: str = 'raise') -> torch.Tensor:
    """Global row indices of rank `rank`'s share of a batch with the reference's layout -- nerf rows, then sparse-depth
    rows (select_batch_indices, DataPreprocessor01.py:544-563).  Each class is split separately into `world` equal
    contiguous parts, so that every rank holds N_nerf/R + N_sd/R rows: MSE / VisibilityPrior average over the nerf rows
    (MSE01.py:55-59, VisibilityPriorLoss01.py:74-80), SparseDepthMSE over the sparse-depth rows (SparseDepthMSE01.py:59-63)
    and VisibilityLoss over all rows -- with equal per-class counts on every rank the mean of the rank means IS the global
    mean and averaging the gradients is exact.

    `batch['row_class_counts'] = (n_nerf, n_sparse_depth)` (host integers; the batch builder knows them: nerf rows first, then the
    sparse-depth rows) lets the ids be computed on the host without looking at the device masks -- no stream synchronisation per
    iteration.  uneven='trim': a class whose count does not divide (the short last batch of an epoch, the arbitrary number of
    sparse-depth pixels of a scene) loses its last count % world rows instead of raising."""
    n = batch['rays_o'].shape[0]
    dev = batch['rays_o'].device
    counts = batch.get('row_class_counts')
    if counts is not None:
        n_nerf, n_sd = int(counts[0]), int(counts[1])
        if n_nerf + n_sd != n:
            raise ValueError(f'row_class_counts {tuple(counts)} do not add up to the {n} rows of the batch')
        parts, start = [], 0
        for cnt in (n_nerf, n_sd):
            s = shard_rows(cnt, rank, world, uneven)
            parts.append(torch.arange(start + s.start, start + s.stop, device=dev))
            start += cnt
        return torch.cat(parts)
    m_nerf = batch.get('indices_mask_nerf')
    m_sd = batch.get('indices_mask_sparse_depth')
    if m_sd is None and (m_nerf is None or bool(m_nerf.all())):
        s = shard_rows(n, rank, world, uneven)
        return torch.arange(s.start, s.stop, device=dev)
    m_nerf = m_nerf.bool() if m_nerf is not None else torch.ones(n, dtype=torch.bool, device=dev)
    m_sd = m_sd.bool() if m_sd is not None else torch.zeros(n, dtype=torch.bool, device=dev)
    parts = []
    for name, m in (('nerf', m_nerf & ~m_sd), ('sparse-depth', m_sd), ('unclassified', ~m_nerf & ~m_sd)):
        ids = torch.nonzero(m, as_tuple=False)[:, 0]
        if ids.numel() % world and uneven != 'trim':
            raise ValueError(f'{ids.numel()} {name} rows do not split evenly over {world} ranks')
        per = ids.numel() // world
        parts.append(ids[rank * per:(rank + 1) * per])
    return torch.cat(parts)


def shard_batch(batch: dict, rank: int, world: int, uneven: str = 'raise') -> dict:
    """This rank's shard of a global batch dict (row-class aware, see shard_row_ids).  Every tensor whose first dimension is
    the row count is indexed; `common_data` and scalars are shared.  `rng_ray_ids` carries the rows' global indices so
    that the on-device Philox streams of the R shards are exactly the streams one process would draw for the whole batch."""
    n = batch['rays_o'].shape[0]
    ids = shard_row_ids(batch, rank, world, uneven)
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n and k != 'common_data':
            out[k] = v[ids.to(v.device)]
        elif k == 'common_data':
            out[k] = dict(v)
        else:
            out[k] = v
    if batch.get('row_class_counts') is not None:
        out['row_class_counts'] = tuple(int(c) // world for c in batch['row_class_counts'])
    out['rng_ray_ids'] = ids if 'rng_ray_ids' not in batch else batch['rng_ray_ids'][ids.to(batch['rng_ray_ids'].device)]
    return out


# ------------------------------------------------------------------------------------------------ self-verification of a sharded step
def verify_sharded_gradient(grad_fn, global_batch: dict, rank: int, world: int) -> dict:
    """Parity evidence for a multi-rank step, computed by the ranks themselves (bench.py --gpus N, tests/test_hip_multigpu.py): every rank
    takes ITS shard of `global_batch` (shard_batch: row-class aware, Philox streams keyed by the rows' global indices), all-reduces the
    shard gradients with the step's own collective (all_reduce_mean_flat), and ALSO computes the gradient of the WHOLE global batch
    locally; the two must agree -- what torch.nn.DataParallel's gather + one backward gives the reference for free (reference
    src/Trainer01.py:517; the losses are means over their row classes, loss_functions/MSE01.py:55-59).

    grad_fn(batch) -> flat fp32 gradient (any device) of the batch's TotalLoss at the current weights, NO collective inside; it must draw
    the same random numbers for a row whatever batch the row arrives in.  -> {'rel_l2': max over ranks of |reduced - whole| / |whole|,
    'rel_l2_rank': this rank's, 'whole_norm': |whole|, 'ranks': ranks that took part}.  World size 1 (forced collectives): the shard is
    the batch, the collective is RCCL's single-rank all-reduce."""
    shard = shard_batch(global_batch, rank, world)
    g = grad_fn(shard).detach().clone()
    all_reduce_mean_flat(g)
    whole = grad_fn(global_batch).detach()
    den = whole.double().norm()
    rel = ((g.double() - whole.double()).norm() / den.clamp_min(1e-300)).reshape(1)
    worst, ranks = rel.clone(), torch.ones(1, dtype=torch.float64, device=rel.device)
    if _active():
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
    return {'rel_l2': float(worst.item()), 'rel_l2_rank': float(rel.item()), 'whole_norm': float(den.item()), 'ranks': int(ranks.item())}


def params_identical(params: Iterable[torch.Tensor]) -> dict:
    """Are the ranks' parameters the same BITS (identical weights, identical reduced gradient, identical Adam step on every rank)?  Two
    checksums per rank -- the sum of the parameters' bit patterns as integers (exact, order independent) and their fp64 sum -- are compared
    with rank 0's (broadcast) and the differences reduced with MAX.  -> {'identical': bool, 'max_bits_diff': int, 'max_sum_diff': float}."""
    ps = [p.detach() for p in params]
    bits = torch.stack([p.contiguous().view(torch.int32).to(torch.int64).sum() for p in ps]).sum().reshape(1)
    fsum = torch.stack([p.double().sum() for p in ps]).sum().reshape(1)
    d_bits, d_sum = torch.zeros(1, dtype=torch.float64, device=bits.device), torch.zeros(1, dtype=torch.float64, device=bits.device)
    if _active():
        b0, f0 = bits.clone(), fsum.clone()
        dist.broadcast(b0, src=0)
        dist.broadcast(f0, src=0)
        d_bits, d_sum = (bits - b0).abs().double(), (fsum - f0).abs()
        dist.all_reduce(d_bits, op=dist.ReduceOp.MAX)
        dist.all_reduce(d_sum, op=dist.ReduceOp.MAX)
    return {'identical': bool(d_bits.item() == 0 and d_sum.item() == 0), 'max_bits_diff': int(d_bits.item()), 'max_sum_diff': float(d_sum.item())}
