"""Counter-based RNG shared by host and device: Philox4x32-10.

The reference never seeds ``random`` / ``numpy.random``, so "same seed" has no
meaning there; the engine defines its own streams instead.  Every draw is a pure function of
``(seed, global env id, step counter, purpose, index)``, which makes results independent of how
environments are sharded over GPUs and lets the tests predict device draws on the host.
Seed and env id sit in separate Philox words -- key = (seed_lo, seed_hi), counter = (step,
purpose | index << 8, gid_lo, gid_hi) -- so fleets drawn under two seeds are independent samples
(with key = seed ^ gid, seed s' merely permuted the envs of seed s).

Device twin: ``philox()`` / ``u01()`` in ``csrc/cosim_kernels.hip``.
Purposes: 0 action-delay draw, 1 sensor noise (index = frame element), 2 init-qpos noise
(index = i-th noisy joint), 3 mass noise (host only, index = body id), 4 gain noise (host only).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)

PURPOSE_DELAY, PURPOSE_SENSOR, PURPOSE_INIT, PURPOSE_MASS, PURPOSE_GAIN = 0, 1, 2, 3, 4


def philox4x32(k0, k1, c0, c1, c2, c3):
    """Vectorised Philox4x32-10; all inputs broadcastable uint32 arrays. Returns 4 uint32 arrays."""
    k0, k1, c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in np.broadcast_arrays(k0, k1, c0, c1, c2, c3))
    k0, k1 = k0.copy(), k1.copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * M0
            p1 = c2.astype(np.uint64) * M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = k0 + W0
            k1 = k1 + W1
    return c0, c1, c2, c3


def u01(x):
    """uint32 -> float32 in (0, 1): same mapping as the device ``u01``."""
    x = np.asarray(x, dtype=np.uint32)
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) + np.float32(0.5 / 16777216.0)).astype(np.float32)


def seed_key(seed: int):
    """Philox key words of a 64-bit seed."""
    return np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)


def env_words(env_ids):
    """Counter words 2 and 3: low / high half of the global env id."""
    gid = np.asarray(env_ids, dtype=np.uint64)
    return (gid & MASK32).astype(np.uint32), (gid >> np.uint64(32)).astype(np.uint32)


def uniform(seed: int, env_ids, step, purpose: int, index):
    """First output word of the stream as float32 uniform in (0,1); shape = broadcast(env_ids, step, index)."""
    k0, k1 = seed_key(seed)
    g0, g1 = env_words(env_ids)
    c1 = np.uint32(purpose) | (np.asarray(index, dtype=np.uint32) << np.uint32(8))
    c0, _, _, _ = philox4x32(k0, k1, np.asarray(step, dtype=np.uint32), c1, g0, g1)
    return u01(c0)
