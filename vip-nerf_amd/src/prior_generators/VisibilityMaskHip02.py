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

    # ---------------------------------------------------------------------------------- a whole scene (reference :233-279)
    def generate_scene(self, frames, extrinsics, intrinsics, min_depth: float, max_depth: float, frame_nums=None, output_dirpath=None,
                       keep_on_device: bool = False):
        """Every ordered pair of a scene's training frames, as the reference's start_generation loop does (:248-279), without the
        database plumbing around it: `frames` (n,h,w,3) uint8, `extrinsics` (n,4,4), `intrinsics` (n,3,3).
        -> (masks (n, n-1, h, w) bool, weights (n, n-1, h, w) float64): row f1 holds the other frames in increasing order, the
        array the reference's loader builds (src/data_loaders/NerfLlffDataLoader01.py:131-160) and RayGeneratorHip takes as
        `visibility_prior`.  With `output_dirpath` the pair files are written in the reference's on-disk layout
        (`visibility_masks/{f1:04}_{f2:04}.npy` + `.png` with 255 = visible, `visibility_weights/{f1:04}_{f2:04}.npy` + `.png`
        with round(255 w)), so the reference's own loader reads them.  keep_on_device: return torch tensors on the GPU instead."""
        n = len(frames)
        frame_nums = list(range(n)) if frame_nums is None else [int(x) for x in frame_nums]
        if len(frame_nums) != n or len(extrinsics) != n or len(intrinsics) != n:
            raise L.VipNerfHipError('generate_scene: frames, extrinsics, intrinsics and frame_nums must have one entry per frame')
        masks, weights = [], []
        for i in range(n):
            row_m, row_w = [], []
            for j in range(n):
                if j == i:
                    continue
                w64, mask = self._run(frames[i], frames[j], extrinsics[i], extrinsics[j], intrinsics[i], intrinsics[j], min_depth, max_depth)
                row_w.append(w64)
                row_m.append(mask != 0)
                if output_dirpath is not None:
                    stem = f'{frame_nums[i]:04}_{frame_nums[j]:04}'
                    self.save_mask(Path(output_dirpath) / f'visibility_masks/{stem}.npy', row_m[-1].cpu().numpy(), as_image=True)
                    self.save_weights(Path(output_dirpath) / f'visibility_weights/{stem}.npy', w64.cpu().numpy(), as_png=True)
            masks.append(torch.stack(row_m))
            weights.append(torch.stack(row_w))
        masks, weights = torch.stack(masks), torch.stack(weights)
        if keep_on_device:
            return masks, weights
        return masks.cpu().numpy(), weights.cpu().numpy()

    @staticmethod
    def _write_png(path: Path, image_u8: numpy.ndarray):
        from PIL import Image                     # the reference writes through skimage.io (absent here); same 8-bit greyscale PNG
        Image.fromarray(image_u8, mode='L').save(path.as_posix())

    @classmethod
    def _save(cls, path: Path, array: numpy.ndarray, preview_u8: numpy.ndarray, also_png: bool):
        """One pair file in the reference's formats: `.npy` holds the array itself (optionally with an 8-bit PNG preview beside
        it), `.png` only the preview."""
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        kind = path.suffix.lower()
        if kind == '.npy':
            numpy.save(path.as_posix(), array)
            if also_png:
                cls._write_png(path.with_suffix('.png'), preview_u8)
        elif kind == '.png':
            cls._write_png(path, preview_u8)
        else:
            raise RuntimeError(f'Unknown format: {path.as_posix()}')

    @classmethod
    def save_mask(cls, path: Path, mask: numpy.ndarray, as_image: bool = False):
        """Bool mask; preview 255 = visible, 0 = not (what the reference's loader compares against; reference :184-197)."""
        cls._save(path, mask, numpy.where(mask, 255, 0).astype(numpy.uint8), as_image)

    @classmethod
    def save_weights(cls, path: Path, weights: numpy.ndarray, as_png: bool = False):
        """float64 weights in [0, 1]; preview round(255 w) (reference :199-211)."""
        cls._save(path, weights, numpy.rint(weights * 255).astype(numpy.uint8), as_png)


def load_scene_masks(masks_dirpath, frame_nums) -> numpy.ndarray:
    """What the reference's loader does with a scene's mask files (NerfLlffDataLoader01.py:131-143, read_mask :169-177):
    `{f1:04}_{f2:04}.png` == 255 -> (n, n-1, h, w) bool, row f1 = the other frames in the order of frame_nums."""
    from PIL import Image
    masks = []
    for f1 in frame_nums:
        row = []
        for f2 in [x for x in frame_nums if x != f1]:
            path = Path(masks_dirpath) / f'visibility_masks/{int(f1):04}_{int(f2):04}.png'
            row.append(numpy.asarray(Image.open(path.as_posix())) == 255)
        masks.append(row)
    return numpy.array(masks)
