"""On-device ray generation and batch assembly and frame post-processing (row f-2).

Mirrors the parts of the reference's DataPreprocessor (src/data_preprocessors/DataPreprocessor01.py) that sit
immediately before and after the hot path:
  get_next_batch(iter_num)        <- load_cached_next_batch :498-529 (+ :566-615, :702-724) for given ray indices
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


class RayGeneratorHip:
    def __init__(self, resolution, intrinsics, poses, near, far, ndc: bool, device, near_ndc=0.0, far_ndc=1.0,
                 images: torch.Tensor = None, visibility_prior: torch.Tensor = None):
        """intrinsics (n,3,3), poses (n,4,4) (already normalised, float32 as the reference keeps them); images
        (n,h,w,3) float32 in [0,1]; visibility_prior (n,n-1,h,w) float32 (masks or weights)."""
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

    def _generate(self, n_rays, indices=None, first_index=0, want_targets=False, want_o2=False):
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
        rb = L.RayBatch()
        names = {'visibility_prior_masks': 'prior'}
        for k, t in b.items():
            setattr(rb, names.get(k, k), t.data_ptr())
        if n_rays > 0:
            L.check(lib.vipnerf_generate_rays(C.byref(g), n_rays, C.byref(rb), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    'vipnerf_generate_rays')
        return b

    # ---- training side -----------------------------------------------------------------------------------
    def get_next_batch(self, iter_num: int, indices: torch.Tensor):
        """indices: flat ray ids (frame*h*w + y*w + x), e.g. a slice of the reference's shuffled index array."""
        b = self._generate(indices.shape[0], indices=indices, want_targets=True)
        b['iter_num'] = iter_num
        b['num_frames'] = self.n
        b['indices'] = indices
        b['indices_mask_nerf'] = torch.ones(indices.shape[0], dtype=torch.bool, device=self.device)
        b['common_data'] = {'poses': self.poses[None]}
        return b

    # ---- inference side -----------------------------------------------------------------------------------
    def create_test_data(self, frame: int = 0, secondary: bool = False):
        """All h*w rays of camera `frame` (the cameras given at construction)."""
        hw = self.h * self.w
        b = self._generate(hw, first_index=frame * hw, want_o2=secondary)
        b['num_frames'] = self.n
        return b

    def retrieve_inference_outputs(self, out: dict, fine: bool = True):
        lib = L.load()
        sfx = '_fine' if fine else '_coarse'
        hw = self.h * self.w
        dev = self.device
        image = torch.empty(self.h, self.w, 3, dtype=torch.uint8, device=dev)
        res = {'image': image, 'depth': torch.empty(self.h, self.w, device=dev), 'depth_var': torch.empty(self.h, self.w, device=dev)}
        p = lambda t: ops.f32c(t).data_ptr() if t is not None else None
        dn, dvn = out.get(f'depth_ndc{sfx}'), out.get(f'depth_var_ndc{sfx}')
        if self.ndc:
            res['depth_ndc'] = torch.empty(self.h, self.w, device=dev)
            res['depth_var_ndc'] = torch.empty(self.h, self.w, device=dev)
        keep = [ops.f32c(out[f'rgb{sfx}']), ops.f32c(out[f'depth{sfx}']), ops.f32c(out[f'depth_var{sfx}']),
                ops.f32c(dn) if dn is not None else None, ops.f32c(dvn) if dvn is not None else None]
        L.check(lib.vipnerf_postprocess_frame(hw, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(),
                                              keep[3].data_ptr() if keep[3] is not None else None,
                                              keep[4].data_ptr() if keep[4] is not None else None,
                                              image.data_ptr(), res['depth'].data_ptr(), res['depth_var'].data_ptr(),
                                              res['depth_ndc'].data_ptr() if self.ndc else None,
                                              res['depth_var_ndc'].data_ptr() if self.ndc else None,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'vipnerf_postprocess_frame')
        if f'visibility2{sfx}' in out:
            res['visibility2'] = out[f'visibility2{sfx}'].reshape(self.h, self.w, -1).permute(2, 0, 1).contiguous()
        return res


def predict_frame(model, gen: RayGeneratorHip, frame: int = 0, secondary: bool = False):
    """NerfTester.predict_frame (reference src/Tester01.py:57-66): camera -> rays -> eval render -> images, all on
    the GPU."""
    b = gen.create_test_data(frame, secondary)
    with torch.no_grad():
        out = model(b, sec_views_vis=secondary)
    return gen.retrieve_inference_outputs(out, fine=getattr(model, 'fine_mlp_needed', True))
