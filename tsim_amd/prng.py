"""Host-side Threefry-2x32 key handling (the part of ``jax.random`` that stays on the host).

The sampler owns a key that is split **once per batch on the host**
(reference: src/tsim/sampler.py:198,272,399,482); only the per-output splits
and the per-shot draws happen on the device (``k_keygen`` / ``uniform01`` in
``csrc/tsim_kernels.hip.h``).  Semantics are JAX's ``threefry2x32`` PRNG with
``jax_threefry_partitionable=True`` (the default of the pinned jax versions):

* ``key(seed)``   -> ``(seed >> 32, seed & 0xffffffff)``
* ``split(k)[j]`` -> the output pair of ``threefry2x32(k, counter=(0, j))``

Plain Python integers, no numpy needed: this runs a handful of times per batch.
"""

from __future__ import annotations

_M = 0xFFFFFFFF
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x: int, r: int) -> int:
    return ((x << r) | (x >> (32 - r))) & _M


def threefry2x32(k0: int, k1: int, c0: int, c1: int) -> tuple[int, int]:
    """One Threefry-2x32 block (20 rounds)."""
    ks = (k0 & _M, k1 & _M, (k0 ^ k1 ^ 0x1BD11BDA) & _M)
    x0 = (c0 + ks[0]) & _M
    x1 = (c1 + ks[1]) & _M
    for blk in range(5):
        for r in _ROT[blk & 1]:
            x0 = (x0 + x1) & _M
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(blk + 1) % 3]) & _M
        x1 = (x1 + ks[(blk + 2) % 3] + blk + 1) & _M
    return x0, x1


Key = tuple  # (hi, lo) uint32 pair


def key(seed: int) -> Key:
    """``jax.random.key(seed)`` for the threefry2x32 implementation."""
    seed = int(seed)
    return ((seed >> 32) & _M, seed & _M)


def split(k: Key) -> tuple[Key, Key]:
    """``new_key, subkey = jax.random.split(key)``."""
    return threefry2x32(k[0], k[1], 0, 0), threefry2x32(k[0], k[1], 0, 1)
