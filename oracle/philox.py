"""
Counter-based uniform generator shared by the oracle and the HIP kernels.

TEST INFRASTRUCTURE (oracle).  The reference draws its sampling noise from
``jax.random.uniform`` (render.py:142, 241-247); JAX's threefry stream cannot be
reproduced here (JAX absent, version unpinned — SURVEY.md §7 "RNG"), so the
build defines parity on *identical uniforms*: either explicit ``u`` arrays, or
this Philox4x32-10 generator which the kernels implement bit-for-bit
(learn-nerf_amd/csrc/philox.h).

Element ``e`` of stream ``s`` under 64-bit ``seed``:
    counter = (lo32(e>>2), hi32(e>>2), s, 0), key = (lo32(seed), hi32(seed))
    word    = philox4x32_10(counter, key)[e & 3]
    u       = (word >> 8) * 2**-24          in [0, 1)
"""

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011). All args uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32)
    c3 = np.asarray(c3, dtype=np.uint32)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def uniform(seed: int, stream: int, first: int, count: int) -> np.ndarray:
    """float32 uniforms for elements first .. first+count-1 of ``stream``."""
    e = np.arange(first, first + count, dtype=np.uint64)
    ctr = e >> np.uint64(2)
    c0 = (ctr & _MASK).astype(np.uint32)
    c1 = (ctr >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(count, stream, dtype=np.uint32)
    c3 = np.zeros(count, dtype=np.uint32)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    out = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, seed >> 32)
    words = np.stack(out, axis=-1)  # [count, 4]
    sel = (e & np.uint64(3)).astype(np.int64)
    w = words[np.arange(count), sel]
    return ((w >> np.uint32(8)).astype(np.float32)) * np.float32(2.0 ** -24)


def ray_uniforms(seed: int, stream: int, ray_offset: int, n_rays: int, count: int) -> np.ndarray:
    """[n_rays, count] uniforms; element index = (ray_offset + n) * count + i."""
    if count == 0:
        return np.zeros((n_rays, 0), dtype=np.float32)
    return uniform(seed, stream, ray_offset * count, n_rays * count).reshape(n_rays, count)
