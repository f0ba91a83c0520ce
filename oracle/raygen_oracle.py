"""CPU oracle for the callers' sides of the hot path.  TEST INFRASTRUCTURE ONLY.

numpy float32 restatement of the reference's ray generation and inference post-processing:
  get_rays                  src/data_preprocessors/DataPreprocessor01.py:335-353
  get_ndc_rays              src/data_preprocessors/DataPreprocessor01.py:354-373
  get_view_dirs             src/data_preprocessors/DataPreprocessor01.py:375-378
  secondary camera centres  src/models/VipNeRF01.py:88-98
  post_process_image/depth  src/data_preprocessors/DataPreprocessor01.py:1074-1084
Pinned against the reference by tests/golden/f6_raygen.npz (oracle/gen_golden.py imports DataPreprocessor01 with an
empty `skimage` stub -- skimage is absent here and is only used for optional down-scaling).
"""
import numpy as np

f32 = np.float32


def camera_tables(intrinsic, pose, resolution):
    """-> (kinv (3,3) f32, pose (3,4) f32, ndc_cx, ndc_cy): what the device kernel consumes.  The NDC coefficients
    are evaluated exactly as the reference's expression does (`-1. / (w / (2. * fx))` with fx a float32 scalar)."""
    h, w = resolution
    K = np.asarray(intrinsic, dtype=f32)
    kinv = np.linalg.inv(K).astype(f32)
    fx, fy = K[0, 0], K[1, 1]
    cx = f32(-1. / (w / (2. * fx)))
    cy = f32(-1. / (h / (2. * fy)))
    return kinv, np.asarray(pose, dtype=f32)[:3, :4].copy(), cx, cy


def get_rays(resolution, intrinsic, pose):
    h, w = resolution
    kinv, p, _, _ = camera_tables(intrinsic, pose, resolution)
    x, y = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32), indexing='xy')
    dirs = np.stack([(kinv[i, 0] * x + kinv[i, 1] * y) + kinv[i, 2] for i in range(3)], axis=-1).astype(f32)
    dirs[..., 1:] *= -1
    rays_d = np.stack([(dirs[..., 0] * p[i, 0] + dirs[..., 1] * p[i, 1]) + dirs[..., 2] * p[i, 2] for i in range(3)], axis=-1)
    rays_o = np.broadcast_to(p[:, 3], rays_d.shape)
    return rays_o.astype(f32), rays_d.astype(f32)


def get_view_dirs(rays_d):
    n = np.sqrt((rays_d[..., 0] * rays_d[..., 0] + rays_d[..., 1] * rays_d[..., 1]) + rays_d[..., 2] * rays_d[..., 2])
    return (rays_d / n[..., None]).astype(f32)


def get_ndc_rays(rays_o, rays_d, resolution, intrinsic, near):
    _, _, cx, cy = camera_tables(intrinsic, np.eye(4), resolution)
    near32, two_near = f32(near), f32(2. * near)
    t = -(near32 + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    o_ndc = np.stack([cx * o[..., 0] / o[..., 2], cy * o[..., 1] / o[..., 2], f32(1.) + two_near / o[..., 2]], -1)
    d_ndc = np.stack([cx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2]),
                      cy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2]), -two_near / o[..., 2]], -1)
    return o_ndc.astype(f32), d_ndc.astype(f32)


def secondary_origins(poses, image_id):
    nf = poses.shape[0]
    cols = []
    for i in range(nf - 1):
        other = i + (i >= image_id).astype(np.int64)
        cols.append(poses[other][:, :3, 3])
    return np.stack(cols, axis=1).astype(f32)


def post_process_image(rgb):
    return np.round(np.clip(rgb, 0, 1) * 255).astype(np.uint8)


def post_process_depth(depth):
    return np.clip(depth, 0, np.inf).astype(f32)
