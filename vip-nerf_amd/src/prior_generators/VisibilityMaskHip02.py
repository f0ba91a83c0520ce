"""GPU drop-in for the reference's visibility-prior generator
(src/prior_generators/visibility/VisibilityMask02_NeRF_LLFF.py: class VisibilityWeightsComputer): same
`compute_weights(frame1, frame2, extrinsic1, extrinsic2, intrinsic1, intrinsic2, min_depth, max_depth)` signature and
return value (float64 (h,w) weights on the host); `compute_masks` adds the `weights > 0.5` step of :275-279.
The per-pixel plane sweep runs in one HIP kernel (vipnerf_visibility_prior); the 4x4 / 3x3 matrix algebra and the
inverse-depth plane list are evaluated on the host with numpy exactly as the reference writes them."""
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


class VisibilityWeightsComputerHip:
    def __init__(self, configs: dict, device='cuda:0'):
        self.configs = configs
        self.device = torch.device(device)

    @staticmethod
    def get_depth_planes(min_depth, max_depth, num_depth_planes):
        return 1 / numpy.linspace(1 / min_depth, 1 / max_depth, num_depth_planes)

    def _run(self, frame1, frame2, extrinsic1, extrinsic2, intrinsic1, intrinsic2, min_depth, max_depth):
        lib = L.load()
        if frame1.dtype != numpy.uint8 or frame2.dtype != numpy.uint8:
            raise L.VipNerfHipError('frames must be uint8 (h,w,3) images, as read from the dataset PNGs')
        h, w = frame1.shape[:2]
        if intrinsic2 is None:
            intrinsic2 = numpy.copy(intrinsic1)
        planes = self.get_depth_planes(min_depth, max_depth, self.configs['num_depth_planes']).astype(numpy.float64)
        transformation = numpy.matmul(extrinsic2, numpy.linalg.inv(extrinsic1))
        p = L.Psv()
        p.height, p.width, p.n_planes = h, w, planes.size
        p.k1_inv[:] = numpy.linalg.inv(intrinsic1).astype(numpy.float64).reshape(-1).tolist()
        p.transform[:] = numpy.asarray(transformation, numpy.float64)[:3, :4].reshape(-1).tolist()
        p.k2[:] = numpy.asarray(intrinsic2, numpy.float64).reshape(-1).tolist()
        p.temperature = float(self.configs['temperature'])
        d_planes = torch.from_numpy(planes).to(self.device)
        f1 = torch.from_numpy(numpy.ascontiguousarray(frame1[:, :, :3])).to(self.device)
        f2 = torch.from_numpy(numpy.ascontiguousarray(frame2[:, :, :3])).to(self.device)
        p.planes, p.frame1, p.frame2 = d_planes.data_ptr(), f1.data_ptr(), f2.data_ptr()
        w64 = torch.empty(h, w, dtype=torch.float64, device=self.device)
        mask = torch.empty(h, w, dtype=torch.uint8, device=self.device)
        with ops.on_device(d_planes, f1, f2, w64, mask) as dev:
            L.check(lib.vipnerf_visibility_prior(C.byref(p), w64.data_ptr(), None, mask.data_ptr(), ops._stream(dev)),
                    'vipnerf_visibility_prior')
        return w64, mask

    def compute_weights(self, frame1, frame2, extrinsic1, extrinsic2, intrinsic1, intrinsic2, min_depth: float, max_depth: float):
        return self._run(frame1, frame2, extrinsic1, extrinsic2, intrinsic1, intrinsic2, min_depth, max_depth)[0].cpu().numpy()

    def compute_masks(self, frame1, frame2, extrinsic1, extrinsic2, intrinsic1, intrinsic2, min_depth: float, max_depth: float):
        w64, mask = self._run(frame1, frame2, extrinsic1, extrinsic2, intrinsic1, intrinsic2, min_depth, max_depth)
        return w64.cpu().numpy(), mask.cpu().numpy().astype(bool)
