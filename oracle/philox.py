"""Philox4x32-10 + Box-Muller: numpy restatement of the counter-based generator behind gldm_denoise_rng.
TEST INFRASTRUCTURE: see oracle/__init__.py.

No reference counterpart: the reference draws the DDPM step noise with torch.randn on the device
(grasp_ldm/models/diffusion/gaussian_diffusion.py:258-272, through the scheduler's step).  The product's in-kernel
generator replaces the *source* of those normals, not their use; this file pins the bits of that generator:

* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), checked against the
  known-answer vectors published with Random123 (tests/test_philox_cpu.py);
* counter = (latent index low word, latent index high word, position // 4, step), key = seed (low, high);
* words (u0, u1) and (u2, u3) -> Box-Muller pairs: a = (u + 1) 2^-32 in (0, 1], angle = u' 2 pi 2^-32,
  z = sqrt(-2 ln a) (cos, sin); the four normals of one counter are positions 4 (l // 4) + 0..3.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [..., 2] uint32 (broadcastable) -> [..., 4] uint32."""
    c = np.asarray(counter, dtype=np.uint64)
    k = np.asarray(key, dtype=np.uint64)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        c1, c3, c0, c2 = p1 & _MASK, p0 & _MASK, n0 & _MASK, n2 & _MASK
        k0 = (k0 + np.uint64(_W0)) & _MASK
        k1 = (k1 + np.uint64(_W1)) & _MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def step_noise(seed, base, step, n, seq_len):
    """[n, seq_len] f32: the unit normals of latents base .. base + n - 1 at `step` (f32 arithmetic like the kernel; the
    kernel's fast log / sin / cos differ in the last bits: compare with 1e-5)."""
    g = np.arange(base, base + n, dtype=np.uint64)[:, None]
    blk = np.arange((seq_len + 3) // 4, dtype=np.uint64)[None, :]
    ctr = np.stack(np.broadcast_arrays(g & _MASK, g >> np.uint64(32), blk, np.uint64(step)), axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    u = philox4x32_10(ctr, key).astype(np.float32)           # exact up to 2^24, rounded above: like (float)u on the device
    inv = np.float32(2.3283064365386963e-10)
    out = np.empty((n, blk.shape[1], 4), dtype=np.float32)
    for h in range(2):
        a = (u[..., 2 * h] + np.float32(1.0)) * inv
        ang = u[..., 2 * h + 1] * (np.float32(6.283185307179586) * inv)
        rad = np.sqrt(np.float32(-2.0) * np.log(a))
        out[..., 2 * h] = rad * np.cos(ang)
        out[..., 2 * h + 1] = rad * np.sin(ang)
    return out.reshape(n, -1)[:, :seq_len]
