"""CPU restatement of the on-device random-number generator.  TEST INFRASTRUCTURE ONLY.

The reference draws its random numbers with torch.rand / torch.randn on the CPU generator inside the ray loop
(src/models/VipNeRF01.py:200 stratified jitter, :242 inverse-CDF draws, :551 sigma noise) and copies them to the device.
The HIP path draws them on the device from Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
as 1, 2, 3", SC'11; the Random123 library) keyed by (seed, offset, stream, global ray index) -- vip-nerf_amd/csrc/
vipnerf_common.h.  The two generators cannot agree number for number (parity tests inject the reference's recorded draws
instead); what IS pinned here is that the device generator is the published Philox4x32-10 (Random123's known-answer
vectors, KAT below) and that the uniform / normal transforms and the keying are what this file restates.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)

# Random123 kat_vectors, "philox4x32 10": counter (4 words), key (2 words) -> output (4 words)
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(counters: np.ndarray, keys: np.ndarray) -> np.ndarray:
    """counters (n,4) uint32, keys (n,2) uint32 -> (n,4) uint32."""
    c = np.asarray(counters, dtype=np.uint32).reshape(-1, 4).copy()
    k = np.asarray(keys, dtype=np.uint32).reshape(-1, 2).copy()
    for _ in range(10):
        p0 = M0 * c[:, 0].astype(np.uint64)
        p1 = M1 * c[:, 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        c = np.stack([hi1 ^ c[:, 1] ^ k[:, 0], lo1, hi0 ^ c[:, 3] ^ k[:, 1], lo0], axis=1)
        with np.errstate(over='ignore'):
            k = np.stack([k[:, 0] + W0, k[:, 1] + W1], axis=1)
    return c


def _words(seed: int, offset: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """Keying of vipnerf_common.h: counter = (idx_lo, idx_hi, stream, offset_lo), key = (seed_lo, seed_hi ^ offset_hi)."""
    idx = np.asarray(idx, dtype=np.uint64).reshape(-1)
    n = idx.shape[0]
    c = np.empty((n, 4), np.uint32)
    c[:, 0] = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c[:, 1] = (idx >> np.uint64(32)).astype(np.uint32)
    c[:, 2] = np.uint32(stream)
    c[:, 3] = np.uint32(offset & 0xFFFFFFFF)
    k = np.empty((n, 2), np.uint32)
    k[:, 0] = np.uint32(seed & 0xFFFFFFFF)
    k[:, 1] = np.uint32(((seed >> 32) ^ (offset >> 32)) & 0xFFFFFFFF)
    return philox4x32_10(c, k)


def rng_uniform(seed: int, offset: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """U[0,1) with 24 bits, like torch.rand's float32."""
    r = _words(seed, offset, stream, idx)
    return ((r[:, 0] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def rng_normal(seed: int, offset: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """N(0,1) by Box-Muller from two 24-bit uniforms (u1 in (0,1], u2 in [0,1)); float64 here, the device evaluates
    logf / cosf in float32, so the comparison carries a tolerance."""
    r = _words(seed, offset, stream, idx)
    u1 = ((r[:, 0] >> np.uint32(8)).astype(np.float64) + 1.0) / 16777216.0
    u2 = (r[:, 1] >> np.uint32(8)).astype(np.float64) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
