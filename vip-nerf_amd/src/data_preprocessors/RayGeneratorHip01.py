"""On-device ray generation and batch assembly and frame post-processing (row f-2).

Mirrors the parts of the reference's DataPreprocessor (src/data_preprocessors/DataPreprocessor01.py) that sit
immediately before and after the hot path:
  BatchIndexScheduler.next(iter)  <- generate_indices :248-265 + select_batch_indices :532-563 (host-side index schedule:
                                     shuffled ray ids, pre-crop, epoch reshuffle, the sparse-depth rows' own schedule)
  get_next_batch(iter_num)        <- load_cached_next_batch :498-529 (+ :566-615 nerf rows, :635-681 sparse-depth rows,
                                     :702-724 visibility prior) for given ray indices
  create_test_data(pose, ...)     <- :776-864 (already pre-processed poses; the pose normalisation of :906-946 is a
                                     once-per-scene host computation and stays on the host)
  retrieve_inference_outputs(out) <- :866-894
The reference caches every ray of every training frame (n*h*w rows of ~30 floats) and gathers rows per iteration
(~20 masked-scatter launches); here the rays of the selected pixels are recomputed on the GPU from the cameras in
ONE launch (vipnerf_generate_rays).  No CPU fallback.
"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy
import torch

try:
    import vipnerf_hip  # noqa: F401
except ImportError:
    for cand in (os.environ.get('VIPNERF_HIP_ROOT'), str(Path(__file__).resolve().parents[2])):
        if cand and cand not in sys.path:
            sys.path.insert(0, cand)
from vipnerf_hip import _lib as L
from vipnerf_hip import ops


class BatchIndexScheduler:
    """The reference's per-iteration ray-index schedule, on the host like the reference's (a few hundred bytes per step):
    a shuffled array of flat ray ids frame*h*w + y*w + x, consumed `num_rays` at a time and reshuffled when exhausted
    (select_batch_indices, DataPreprocessor01.py:532-543), optionally restricted to the central crop for the first
    `precrop_iterations` iterations (generate_indices :248-265), plus -- when sparse depth is on -- a second shuffled
    array of the pixels that have a depth, consumed `num_rays_sparse` at a time and appended (:549-557).

    All shuffles are numpy.random.shuffle on numpy's GLOBAL generator, in the reference's order (ray ids first, then the
    sparse-depth ids, DataPreprocessor01.py:231,239), so that after numpy.random.seed(s) the schedule is the reference's,
    index for index (golden F6b).  Two behaviours of the reference are kept on purpose because they shape the schedule:
    the last batch of an epoch is short (the slice runs off the end), and at iter_num == precrop_iterations the reference
    draws a fresh full-frame permutation but discards it (:536-537 ignores generate_indices' return value), so the
    cropped ids stay in use; `keep_reference_precrop_quirk=False` switches to the full frame there instead."""

    def __init__(self, n_frames: int, h: int, w: int, num_rays: int, precrop_fraction: float = None,
                 precrop_iterations: int = 0, sparse_depths=None, num_rays_sparse: int = 0,
                 keep_reference_precrop_quirk: bool = True):
        self.n, self.h, self.w = int(n_frames), int(h), int(w)
        self.num_rays, self.num_rays_sparse = int(num_rays), int(num_rays_sparse)
        self.precrop_fraction, self.precrop_iterations = precrop_fraction, int(precrop_iterations or 0)
        self.quirk = keep_reference_precrop_quirk
        self.i_batch = self.i_batch_sparse = 0
        self.indices = self._generate(0)
        self.indices_sparse = None
        if sparse_depths is not None:
            d = numpy.asarray(sparse_depths, dtype=numpy.float32).reshape(-1)
            self.indices_sparse = numpy.where(d > 0)[0]
            numpy.random.shuffle(self.indices_sparse)

    def _generate(self, iter_num: int):
        idx = numpy.arange(self.n * self.h * self.w)
        f = self.precrop_fraction
        if f is not None and f < 1 and iter_num < self.precrop_iterations:
            h1, h2 = int(round(self.h / 2 * (1 - f))), int(round(self.h / 2 * (1 + f)))
            w1, w2 = int(round(self.w / 2 * (1 - f))), int(round(self.w / 2 * (1 + f)))
            idx = idx.reshape(self.n, self.h, self.w)[:, h1:h2, w1:w2].ravel()
        numpy.random.shuffle(idx)
        return idx

    def next(self, iter_num: int):
        """-> (flat ray ids of this iteration's rows (int64), row_is_sparse (bool)): nerf rows first, then sparse-depth rows."""
        if self.precrop_fraction is not None and iter_num == self.precrop_iterations:
            fresh = self._generate(iter_num)
            if not self.quirk:
                self.indices, self.i_batch = fresh, 0
        idx = self.indices[self.i_batch:self.i_batch + self.num_rays]
        self.i_batch += self.num_rays
        if self.i_batch >= self.indices.size:
            numpy.random.shuffle(self.indices)
            self.i_batch = 0
        is_sparse = numpy.zeros(idx.shape[0], dtype=bool)
        if self.indices_sparse is not None:
            sd = self.indices_sparse[self.i_batch_sparse:self.i_batch_sparse + self.num_rays_sparse]
            self.i_batch_sparse += self.num_rays_sparse
            if self.i_batch_sparse >= self.indices_sparse.size:
                numpy.random.shuffle(self.indices_sparse)
                self.i_batch_sparse = 0
            idx = numpy.concatenate([idx, sd])
            is_sparse = numpy.concatenate([is_sparse, numpy.ones(sd.shape[0], dtype=bool)])
        return idx.astype(numpy.int64), is_sparse


class RayGeneratorHip:
    def __init__(self, resolution, intrinsics, poses, near, far, ndc: bool, device, near_ndc=0.0, far_ndc=1.0,
                 images: torch.Tensor = None, visibility_prior: torch.Tensor = None, sparse_depths=None, sparse_errors=None,
                 sparse_depths_ndc=None):
        """intrinsics (n,3,3), poses (n,4,4) (already normalised, float32 as the reference keeps them); images
        (n,h,w,3) float32 in [0,1]; visibility_prior (n,n-1,h,w) float32 (masks or weights); sparse_depths / sparse_errors
        (n,h,w) per-pixel sparse depth and reprojection error, -1 where unknown (preprocess_raw_sparse_depth_data,
        DataPreprocessor01.py:161-184); sparse_depths_ndc: computed here like :431-436 when not given."""
        self.h, self.w = int(resolution[0]), int(resolution[1])
        self.ndc, self.device = bool(ndc), torch.device(device)
        self.near, self.far, self.near_ndc, self.far_ndc = float(near), float(far), float(near_ndc), float(far_ndc)
        intrinsics = numpy.asarray(intrinsics, dtype=numpy.float32).reshape(-1, 3, 3)
        self.poses_np = numpy.asarray(poses, dtype=numpy.float32).reshape(-1, 4, 4)
        self.n = self.poses_np.shape[0]
        if intrinsics.shape[0] == 1 and self.n > 1:
            intrinsics = numpy.repeat(intrinsics, self.n, axis=0)
        cams = numpy.zeros((self.n, 25), dtype=numpy.float32)       # vipnerf_camera: 9 + 12 + 2 + 2 pad
        for i in range(self.n):
            K = intrinsics[i]
            cams[i, 0:9] = numpy.linalg.inv(K).astype(numpy.float32).reshape(-1)
            cams[i, 9:21] = self.poses_np[i, :3, :4].reshape(-1)
            fx, fy = K[0, 0], K[1, 1]
            cams[i, 21] = numpy.float32(-1. / (self.w / (2. * fx)))      # same expression, same float32 evaluation
            cams[i, 22] = numpy.float32(-1. / (self.h / (2. * fy)))
        assert C.sizeof(L.Camera) == 25 * 4
        self.cameras = torch.from_numpy(cams).to(self.device)
        self.poses = torch.from_numpy(self.poses_np).to(self.device)
        self.images = ops.f32c(images.to(self.device)) if images is not None else None
        self.prior = ops.f32c(visibility_prior.to(self.device)) if visibility_prior is not None else None
        self.sparse_depths = self.sparse_errors = self.sparse_depths_ndc = None
        if sparse_depths is not None:
            sd = numpy.asarray(sparse_depths, dtype=numpy.float32).reshape(-1)
            self.sparse_depths = torch.from_numpy(sd).to(self.device)
            if sparse_errors is not None:
                self.sparse_errors = torch.from_numpy(numpy.asarray(sparse_errors, dtype=numpy.float32).reshape(-1).copy()).to(self.device)
            if self.ndc:
                if sparse_depths_ndc is None:
                    sparse_depths_ndc = self._depths_to_ndc(sd)
                self.sparse_depths_ndc = torch.from_numpy(numpy.asarray(sparse_depths_ndc, dtype=numpy.float32).reshape(-1).copy()).to(self.device)

    def _depths_to_ndc(self, depths: numpy.ndarray) -> numpy.ndarray:
        """preprocess_sparse_depth_data / convert_depth_to_ndc (DataPreprocessor01.py:431-446) -- once per scene, on the
        host in numpy float32 exactly as the reference evaluates it (near is hard-coded to 1 there), from the rays of
        all n*h*w pixels as the device kernel generates them (bit-identical to the reference's ray cache)."""
        full = self._generate(self.n * self.h * self.w)
        oz = full['rays_o'][:, 2:].cpu().numpy()
        dz = full['rays_d'][:, 2:].cpu().numpy()
        d = depths.reshape(-1, 1)
        tn = -(1 + oz) / dz
        oz_prime = oz + tn * dz
        out = 1 - oz_prime / (oz_prime + (d - tn) * dz)
        out[d == -1] = -1
        return out.astype(numpy.float32)

    def _generate(self, n_rays, indices=None, first_index=0, want_targets=False, want_o2=False, row_is_sparse=None):
        lib = L.load()
        dev = self.device
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        b = {'rays_o': e(n_rays, 3), 'rays_d': e(n_rays, 3), 'view_dirs': e(n_rays, 3), 'near': e(n_rays, 1), 'far': e(n_rays, 1),
             'pixel_id': torch.empty(n_rays, 3, dtype=torch.int32, device=dev)}
        if self.ndc:
            b.update(rays_o_ndc=e(n_rays, 3), rays_d_ndc=e(n_rays, 3), near_ndc=e(n_rays, 1), far_ndc=e(n_rays, 1))
        if want_targets and self.images is not None:
            b['target_rgb'] = e(n_rays, 3)
        if want_targets and self.prior is not None:
            b['visibility_prior_masks'] = e(n_rays, self.n - 1)
        if want_o2 and self.n > 1:
            b['rays_o2'] = e(n_rays, self.n - 1, 3)
        if row_is_sparse is not None:
            b['sparse_depth_values'] = e(n_rays, 1)
            b['sparse_depth_errors'] = e(n_rays, 1)
            if self.ndc:
                b['sparse_depth_values_ndc'] = e(n_rays, 1)
        g = L.RayGen()
        g.height, g.width, g.n_frames, g.ndc = self.h, self.w, self.n, int(self.ndc)
        g.near, g.far, g.near_ndc, g.far_ndc = self.near, self.far, self.near_ndc, self.far_ndc
        g.cameras = self.cameras.data_ptr()
        if indices is not None:
            indices = indices.to(device=dev, dtype=torch.int64).contiguous()
            g.indices = indices.data_ptr()
        g.first_index = int(first_index)
        g.images = self.images.data_ptr() if self.images is not None else None
        g.prior = self.prior.data_ptr() if self.prior is not None else None
        if row_is_sparse is not None:
            row_is_sparse = row_is_sparse.to(device=dev, dtype=torch.uint8).contiguous()
            g.row_is_sparse = row_is_sparse.data_ptr()
            g.sparse_depths = self.sparse_depths.data_ptr() if self.sparse_depths is not None else None
            g.sparse_errors = self.sparse_errors.data_ptr() if self.sparse_errors is not None else None
            g.sparse_depths_ndc = self.sparse_depths_ndc.data_ptr() if self.sparse_depths_ndc is not None else None
        rb = L.RayBatch()
        names = {'visibility_prior_masks': 'prior'}
        for k, t in b.items():
            setattr(rb, names.get(k, k), t.data_ptr())
        if n_rays > 0:
            with ops.on_device(*b.values(), self.cameras, indices, row_is_sparse) as d:
                L.check(lib.vipnerf_generate_rays(C.byref(g), n_rays, C.byref(rb), ops._stream(d)), 'vipnerf_generate_rays')
        return b

    # ---- training side -----------------------------------------------------------------------------------
    def get_next_batch(self, iter_num: int, indices=None, row_is_sparse=None, scheduler: 'BatchIndexScheduler' = None):
        """The training batch dict of load_cached_next_batch (DataPreprocessor01.py:498-529).  indices: flat ray ids
        (frame*h*w + y*w + x) of the rows, nerf rows first; row_is_sparse: bool per row, True = sparse-depth row (:549-563);
        or pass a BatchIndexScheduler and both come from its next(iter_num)."""
        if indices is None:
            if scheduler is None:
                raise L.VipNerfHipError('get_next_batch needs either ray indices or a BatchIndexScheduler')
            indices, row_is_sparse = scheduler.next(iter_num)
        # host-side row-class counts (nerf rows first, then the sparse-depth rows): what vipnerf_hip.dist.shard_row_ids needs to cut a
        # batch into per-rank shards without reading the device masks back
        counts = None
        if not (isinstance(indices, torch.Tensor) and indices.is_cuda) and not (isinstance(row_is_sparse, torch.Tensor) and row_is_sparse.is_cuda):
            n_rows = int(len(indices))
            flags = numpy.asarray(row_is_sparse, dtype=bool) if row_is_sparse is not None else numpy.zeros(0, dtype=bool)
            n_sd = int(numpy.count_nonzero(flags))
            # the counts promise "nerf rows first, then the sparse-depth rows" (what BatchIndexScheduler produces and dist.shard_row_ids'
            # host path assumes); a caller's own interleaved rows get no counts and are sharded by the device masks instead
            if n_sd == 0 or not flags[:n_rows - n_sd].any():
                counts = (n_rows - n_sd, n_sd)
        indices = self._upload(indices, torch.int64)
        sparse_on = self.sparse_depths is not None
        if row_is_sparse is None and sparse_on:
            row_is_sparse = torch.zeros(indices.shape[0], dtype=torch.bool, device=self.device)
        if row_is_sparse is not None:
            row_is_sparse = self._upload(row_is_sparse, torch.bool)
        b = self._generate(indices.shape[0], indices=indices, want_targets=True, row_is_sparse=row_is_sparse if sparse_on else None)
        b['iter_num'] = iter_num
        b['num_frames'] = self.n
        b['indices'] = indices
        if row_is_sparse is not None and sparse_on:
            b['indices_mask_nerf'] = ~row_is_sparse
            b['indices_mask_sparse_depth'] = row_is_sparse
        else:
            b['indices_mask_nerf'] = torch.ones(indices.shape[0], dtype=torch.bool, device=self.device)
        b['common_data'] = {'poses': self.poses[None]}
        if counts is not None and (sparse_on or counts[1] == 0):
            b['row_class_counts'] = counts
        return b

    def _upload(self, x, dtype):
        """Host index / flag arrays -> device WITHOUT draining the stream: a copy from pageable host memory waits for everything
        queued before it (one full stall of the GPU per training iteration); from a pinned staging buffer it is just another
        stream-ordered operation.  A ring of staging buffers, each reused only after the copy that last read it has completed."""
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return x.to(self.device, dtype)
        t = torch.from_numpy(numpy.ascontiguousarray(x)) if isinstance(x, numpy.ndarray) else x
        t = t.to(dtype)
        n = t.numel()
        if n == 0 or self.device.type != 'cuda':
            return t.to(self.device)
        ring = self.__dict__.setdefault('_staging', {})
        slots = ring.setdefault(dtype, [])
        k = self.__dict__.get('_staging_next', 0)
        self._staging_next = k + 1
        i = k % 4
        while len(slots) <= i:
            slots.append([None, None])
        buf, ev = slots[i]
        if ev is not None:
            ev.synchronize()                                   # the copy issued four uploads ago: long done
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 8192), dtype=dtype).pin_memory()
        buf[:n].copy_(t.reshape(-1))
        out = buf[:n].to(self.device, non_blocking=True).reshape(t.shape)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slots[i] = [buf, ev]
        return out

    # ---- inference side -----------------------------------------------------------------------------------
    def create_test_data(self, frame: int = 0, secondary: bool = False, rows=None):
        """All h*w rays of camera `frame` (the cameras given at construction); rows = (r0, r1): the strip of image rows r0 .. r1-1
        only (rays are independent: a frame splits into strips across GPUs with no exchange)."""
        r0, r1 = (0, self.h) if rows is None else (int(rows[0]), int(rows[1]))
        if not 0 <= r0 <= r1 <= self.h:
            raise L.VipNerfHipError(f'create_test_data: rows {rows} outside the {self.h}-row frame')
        b = self._generate((r1 - r0) * self.w, first_index=frame * self.h * self.w + r0 * self.w, want_o2=secondary)
        b['num_frames'] = self.n
        return b

    def retrieve_inference_outputs(self, out: dict, fine: bool = True, rows=None):
        lib = L.load()
        sfx = '_fine' if fine else '_coarse'
        h = self.h if rows is None else int(rows[1]) - int(rows[0])
        hw = h * self.w
        dev = self.device
        image = torch.empty(h, self.w, 3, dtype=torch.uint8, device=dev)
        res = {'image': image, 'depth': torch.empty(h, self.w, device=dev), 'depth_var': torch.empty(h, self.w, device=dev)}
        dn, dvn = out.get(f'depth_ndc{sfx}'), out.get(f'depth_var_ndc{sfx}')
        if self.ndc:
            res['depth_ndc'] = torch.empty(h, self.w, device=dev)
            res['depth_var_ndc'] = torch.empty(h, self.w, device=dev)
        keep = [ops.f32c(out[f'rgb{sfx}']), ops.f32c(out[f'depth{sfx}']), ops.f32c(out[f'depth_var{sfx}']),
                ops.f32c(dn) if dn is not None else None, ops.f32c(dvn) if dvn is not None else None]
        with ops.on_device(*keep, image):
            L.check(lib.vipnerf_postprocess_frame(hw, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(),
                                                  keep[3].data_ptr() if keep[3] is not None else None,
                                                  keep[4].data_ptr() if keep[4] is not None else None,
                                                  image.data_ptr(), res['depth'].data_ptr(), res['depth_var'].data_ptr(),
                                                  res['depth_ndc'].data_ptr() if self.ndc else None,
                                                  res['depth_var_ndc'].data_ptr() if self.ndc else None,
                                                  ops._stream(dev)), 'vipnerf_postprocess_frame')
        if f'visibility2{sfx}' in out:
            v2 = out[f'visibility2{sfx}']
            res['visibility2'] = v2.reshape(h, self.w, v2.shape[-1]).permute(2, 0, 1).contiguous()      # (an empty strip has no -1 to infer)
        return res


def predict_frame(model, gen: RayGeneratorHip, frame: int = 0, secondary: bool = False, rows=None):
    """NerfTester.predict_frame (reference src/Tester01.py:57-66): camera -> rays -> eval render -> images, all on
    the GPU.  rows = (r0, r1): that strip of the frame only (predict_frame_sharded)."""
    b = gen.create_test_data(frame, secondary, rows=rows)
    with torch.no_grad():
        out = model(b, sec_views_vis=secondary)
    return gen.retrieve_inference_outputs(out, fine=getattr(model, 'fine_mlp_needed', True), rows=rows)


def frame_strip(h: int, rank: int, world: int):
    """Rows [r0, r1) of an h-row frame that rank `rank` of `world` renders: contiguous strips, sizes differing by at most one row."""
    return (h * rank) // world, (h * (rank + 1)) // world


def predict_frame_sharded(model, gen: RayGeneratorHip, frame: int = 0, secondary: bool = False, rank: int = 0, world: int = 1,
                          gather: bool = True):
    """One frame on `world` GPUs: every rank renders its strip of rows (no data-path collective -- rays are independent); with
    gather=True the strips are then exchanged (torch.distributed.all_gather_object of the host copies: 2.3 MB of uint8 per
    756 x 1008 frame) and every rank returns the whole frame as host tensors, otherwise its own strip on the GPU."""
    rows = frame_strip(gen.h, rank, world)
    part = predict_frame(model, gen, frame, secondary, rows=rows)
    if not gather or world == 1:
        return part
    import torch.distributed as dist
    host = {k: v.cpu() for k, v in part.items()}
    parts = [None] * world
    dist.all_gather_object(parts, host)
    cat_dim = lambda k: 1 if k == 'visibility2' else 0          # visibility2 is (V, h, w)
    return {k: torch.cat([p[k] for p in parts], dim=cat_dim(k)) for k in host}
