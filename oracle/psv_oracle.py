"""CPU oracle for the visibility-prior generator.  TEST INFRASTRUCTURE ONLY.

numpy float64 restatement of VisibilityWeightsComputer.compute_weights
(reference src/prior_generators/visibility/VisibilityMask02_NeRF_LLFF.py:27-162), per pixel and plane, written from
the math: back-project the pixel at each inverse-depth plane, move it to camera 2, project, sample frame 2
bilinearly with a one-pixel zero border and validity normalisation, channel-mean absolute error against frame 1,
minimum over planes, w = exp(-e/T).  Pinned by tests/golden/f7_visibility_prior.npz (generated from the reference).
"""
import numpy as np


def depth_planes(min_depth, max_depth, n):
    return 1 / np.linspace(1 / min_depth, 1 / max_depth, n)


def compute_weights(frame1, frame2, extrinsic1, extrinsic2, intrinsic1, intrinsic2, min_depth, max_depth, n_planes=64,
                    temperature=10.0):
    h, w = frame1.shape[:2]
    planes = depth_planes(min_depth, max_depth, n_planes)
    T = np.matmul(extrinsic2, np.linalg.inv(extrinsic1))
    K1i = np.linalg.inv(intrinsic1)
    K2 = np.asarray(intrinsic2, np.float64)
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    ray = [(K1i[r, 0] * gx + K1i[r, 1] * gy) + K1i[r, 2] for r in range(3)]
    f2 = np.pad(frame2.astype(np.float32), ((1, 1), (1, 1), (0, 0))).astype(np.float64)
    m2 = np.pad(np.ones((h, w)), ((1, 1), (1, 1)))
    f1 = frame1.astype(np.float64)
    best = np.full((h, w), np.inf)
    for z in planes:
        wp = [z * r for r in ray]
        tw = [((T[r, 0] * wp[0] + T[r, 1] * wp[1]) + T[r, 2] * wp[2]) + T[r, 3] for r in range(3)]
        pr = [(K2[r, 0] * tw[0] + K2[r, 1] * tw[1]) + K2[r, 2] * tw[2] for r in range(3)]
        tx = (pr[0] / pr[2] - gx) + gx
        ty = (pr[1] / pr[2] - gy) + gy
        ox, oy = tx + 1, ty + 1
        fx, cx = np.clip(np.floor(ox), 0, w + 1).astype(int), np.clip(np.ceil(ox), 0, w + 1).astype(int)
        fy, cy = np.clip(np.floor(oy), 0, h + 1).astype(int), np.clip(np.ceil(oy), 0, h + 1).astype(int)
        ox, oy = np.clip(ox, 0, w + 1), np.clip(oy, 0, h + 1)
        wnw = (1 - (oy - fy)) * (1 - (ox - fx))
        wsw = (1 - (cy - oy)) * (1 - (ox - fx))
        wne = (1 - (oy - fy)) * (1 - (cx - ox))
        wse = (1 - (cy - oy)) * (1 - (cx - ox))
        mnw, msw, mne, mse = m2[fy, fx], m2[cy, fx], m2[fy, cx], m2[cy, cx]
        dr = ((wnw * mnw + wsw * msw) + wne * mne) + wse * mse
        err = 0
        for c in range(3):
            nr = ((wnw * f2[fy, fx, c] * mnw + wsw * f2[cy, fx, c] * msw) + wne * f2[fy, cx, c] * mne) + wse * f2[cy, cx, c] * mse
            with np.errstate(invalid='ignore', divide='ignore'):
                warped = np.where(dr > 0, nr / dr, 0)
            err = err + np.abs(warped - f1[..., c])
        best = np.minimum(best, err / 3)
    return np.exp(-best / temperature)
