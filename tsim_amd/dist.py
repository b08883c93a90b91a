"""Shot-parallel sharding of one batch over the GPUs of a node.

The path shards over shots with no data-path exchange: rank ``r`` of ``R``
evaluates the contiguous in-batch rows ``[lo_r, hi_r)`` with ``shot_offset = lo_r``; because the
Threefry counter is the *global* in-batch row index, the gathered result is bit-identical for
every ``R``.  The only collective is the gather of the packed output rows to rank 0 (RCCL over
xGMI with the "nccl" backend; "gloo" on CPU for tests).

``sample_fn(program, f_rows, key, shot_offset) -> uint8/bool[rows, ...]`` is injected: in the
product it is the HIP backend, in the CPU tests (no GPU) it is the oracle - this module contains
no arithmetic of its own.
"""

from __future__ import annotations

import numpy as np


def shard_bounds(B: int, R: int) -> list[tuple[int, int]]:
    """Contiguous, balanced row ranges: the first ``B % R`` ranks get one extra row."""
    if R < 1:
        raise ValueError("R must be >= 1")
    base, extra = divmod(int(B), R)
    bounds, lo = [], 0
    for r in range(R):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def hip_sample_fn(device: int = 0, bit_packed: bool = False):
    """The product ``sample_fn``: one shard through the fused kernel on ``device``."""
    from .backend import get_hip_program

    def fn(program, f_rows, key, shot_offset):
        out, _ = get_hip_program(program, device).sample_batch(
            f_rows, key, shot_offset=shot_offset, bit_packed=bit_packed
        )
        return out

    return fn


def sample_program_sharded(program, f_params: np.ndarray, key, *, sample_fn, group=None, dst: int = 0):
    """Every rank passes the same ``(program, f_params, key)``; rank ``dst`` returns the full
    ``[B, ...]`` result, the others ``None``.  Uses ``torch.distributed`` (already initialised)."""
    import torch
    import torch.distributed as dist

    R = dist.get_world_size(group)
    r = dist.get_rank(group)
    B = int(np.asarray(f_params).shape[0])
    bounds = shard_bounds(B, R)
    lo, hi = bounds[r]
    local = np.ascontiguousarray(sample_fn(program, np.asarray(f_params)[lo:hi], key, lo)).view(np.uint8)
    width = local.shape[1] if local.ndim == 2 else 0
    # equal-size gather: pad every shard to the largest one
    rows = max(b - a for a, b in bounds)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros((rows, width), dtype=torch.uint8, device=dev)
    if hi > lo and width:
        buf[: hi - lo] = torch.from_numpy(local).to(dev)
    gather_list = [torch.empty_like(buf) for _ in range(R)] if r == dst else None
    dist.gather(buf, gather_list, dst=dst, group=group)
    if r != dst:
        return None
    parts = [gather_list[q][: bounds[q][1] - bounds[q][0]].cpu().numpy() for q in range(R)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, width), np.uint8)


def sample_program_multi_device(program, f_params: np.ndarray, key, devices: list[int]) -> np.ndarray:
    """Single-process variant: shard one batch over several GPUs of this process (one handle and
    stream per device, launches overlap, results concatenated on the host)."""
    from .backend import get_hip_program

    f = np.ascontiguousarray(np.asarray(f_params))
    if f.dtype != np.uint8:
        f = (f != 0).astype(np.uint8)
    B, num_f = f.shape
    bounds = shard_bounds(B, len(devices))
    hps = [get_hip_program(program, d) for d in devices]
    n_out = hps[0].num_outputs
    wf, wo = (num_f + 63) // 64, (n_out + 63) // 64
    pending = []
    for hp, (lo, hi) in zip(hps, bounds):
        n = hi - lo
        if n == 0:
            pending.append(None)
            continue
        d_u8 = hp.malloc(n * max(1, num_f))
        d_f = hp.malloc(n * max(1, wf) * 8)
        d_o = hp.malloc(n * wo * 8)
        hp.h2d(d_u8, f[lo:hi])
        hp.pack_bits_device(d_u8.ptr, n, num_f, d_f.ptr)
        hp.sample_batch_device(d_f.ptr, n, num_f, key, d_o.ptr, shot_offset=lo)  # async
        pending.append((hp, n, d_o, (d_u8, d_f)))
    parts = []
    for item in pending:
        if item is None:
            parts.append(np.zeros((0, n_out), np.bool_))
            continue
        hp, n, d_o, _keep = item
        packed = np.zeros((n, wo * 8), np.uint8)
        hp.d2h(packed, d_o)  # synchronises that device's stream
        parts.append(np.unpackbits(packed, axis=1, bitorder="little")[:, :n_out].view(np.bool_))
    return np.concatenate(parts, axis=0)
