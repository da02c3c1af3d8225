"""JAX-compatible frame sampling for the offset phase, without JAX.

``offset_optimization`` draws ``jax.random.permutation(PRNGKey(0), arange(n), independent=True)[:N_SAMPLE_FRAMES]``
(``stac_mjx/compute_stac.py:136-140``).  This restates it: Threefry-2x32 block
cipher (validated against the Random123 known-answer vectors in tests), JAX's ``split`` / ``random_bits``
wiring under ``jax_threefry_partitionable=True`` (jax >= 0.5 default) and the sort-based ``_shuffle``.
The wiring is restated from JAX's published algorithm and could not be checked against a JAX install in
this image; when ``n_frames <= N_SAMPLE_FRAMES`` (BASELINE config 1) every frame is selected and the
permutation only fixes the float32 summation order.
"""

from __future__ import annotations

import math

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M32


def threefry2x32(key, ctr0, ctr1):
    """Threefry-2x32, 20 rounds.  key = (k0, k1); counters are uint32 arrays.  Returns (x0, x1)."""
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    ks = [k0, k1, (k0 ^ k1 ^ np.uint64(0x1BD11BDA)) & _M32]
    x0 = (np.asarray(ctr0, dtype=np.uint64) + ks[0]) & _M32
    x1 = (np.asarray(ctr1, dtype=np.uint64) + ks[1]) & _M32
    rot = ([13, 15, 26, 6], [17, 29, 16, 24])
    for i in range(5):
        for r in rot[i % 2]:
            x0 = (x0 + x1) & _M32
            x1 = _rotl(x1, r)
            x1 = x1 ^ x0
        x0 = (x0 + ks[(i + 1) % 3]) & _M32
        x1 = (x1 + ks[(i + 2) % 3] + np.uint64(i + 1)) & _M32
    return x0.astype(np.uint32), x1.astype(np.uint32)


def prng_key(seed: int):
    """``jax.random.PRNGKey(seed)`` for 0 <= seed < 2**32: key words (0, seed)."""
    return (np.uint32(0), np.uint32(seed & 0xFFFFFFFF))


def split(key, num: int = 2):
    """``jax.random.split`` (partitionable threefry): child i = threefry(key, counter (hi=0, lo=i))."""
    i = np.arange(num, dtype=np.uint32)
    x0, x1 = threefry2x32(key, np.zeros(num, np.uint32), i)
    return [(x0[j], x1[j]) for j in range(num)]


def random_bits32(key, n: int) -> np.ndarray:
    """``jax.random.bits(key, (n,), uint32)`` (partitionable): word i = w0 ^ w1 of threefry(key, (0, i))."""
    i = np.arange(n, dtype=np.uint32)
    x0, x1 = threefry2x32(key, np.zeros(n, np.uint32), i)
    return x0 ^ x1


def permutation(key, n: int) -> np.ndarray:
    """``jax.random.permutation(key, arange(n), independent=True)`` for a 1-D input (``_shuffle``)."""
    x = np.arange(n)
    if n <= 1:
        return x
    rounds = int(math.ceil(3 * math.log(max(1, n)) / math.log(2**32 - 1)))
    for _ in range(rounds):
        key, sub = split(key, 2)
        bits = random_bits32(sub, n)
        x = x[np.argsort(bits, kind="stable")]
    return x


def sample_time_indices(n_frames: int, n_sample_frames: int, seed: int = 0) -> np.ndarray:
    """The frame subset of ``offset_optimization`` (compute_stac.py:136-140)."""
    return permutation(prng_key(seed), n_frames)[:n_sample_frames]
