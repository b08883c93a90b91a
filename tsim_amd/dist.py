"""Shot-parallel sharding over the GPUs of a node, and the one collective that goes with it.

The path shards over shots with no data-path exchange: rank ``r`` of ``R`` evaluates
the contiguous in-batch rows ``[lo_r, hi_r)`` with ``shot_offset = lo_r``; the Threefry counter is the
*global* in-batch row index, so the assembled result is bit-identical for every ``R``.  Only the finished
rows travel: an RCCL gather over xGMI issued by ``libtsim_hip.so`` itself (``tsim_dist_*`` in
``include/tsim_hip.h``) - one process per GPU, **no PyTorch**.  The reference has no multi-device path
(``src/tsim/sampler.py:310`` uses ``jax.devices()[0]``).

* :class:`Communicator` - handle over ``tsim_dist*`` (``ncclCommInitRank`` on this rank's device).
* :func:`rendezvous_tcp` - the 128-byte ``ncclUniqueId`` from rank 0 to every rank over a plain socket.
* :func:`shard_bounds` / :func:`sample_program_sharded` - the sharding logic itself, independent of how rows
  are sampled (``sample_fn``) and gathered (``gather``): the product passes the HIP backend and a
  :class:`Communicator`, the CPU tests the oracle and a ``gloo`` gather.
"""

from __future__ import annotations

import ctypes as C
import os
import socket
import time

import numpy as np

from . import _lib

ID_BYTES = 128
_MAGIC = b"TSIM-RCCL-ID\0"


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


# ---------------------------------------------------------------------------------------------------------
# unique-id exchange
# ---------------------------------------------------------------------------------------------------------


def unique_id() -> bytes:
    """A fresh ``ncclUniqueId`` (call on ONE rank, hand the bytes to all)."""
    buf = (C.c_uint8 * ID_BYTES)()
    _lib.check(_lib.load().tsim_dist_unique_id(buf), "tsim_dist_unique_id")
    return bytes(buf)


def _recv_exact(conn: socket.socket, n: int) -> bytes:
    data = b""
    while len(data) < n:
        chunk = conn.recv(n - len(data))
        if not chunk:
            break
        data += chunk
    return data


def rendezvous_tcp(rank: int, world: int, *, addr: str = "127.0.0.1", port: int, timeout: float = 120.0,
                   make_id=unique_id) -> bytes:
    """Rank 0 creates the id and serves it on ``(addr, port)`` until every other rank has fetched it; the others
    connect (retrying until ``timeout``), announce themselves - magic prefix and their rank - and read it.  Rank 0
    counts DISTINCT ranks: a stray connection (a port scan, a health probe, the second attempt of a rank whose first
    read timed out) is answered or dropped without using up anybody's place, and the listener stays open until all
    ``world - 1`` ranks have been served or the deadline passes.  The reply carries the magic too, so a foreign
    service on the port is recognised and reported instead of being trusted."""
    if world == 1:
        return make_id()
    deadline = time.monotonic() + timeout
    if rank == 0:
        ident = make_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(max(8, 2 * world))
            served: set[int] = set()
            while len(served) < world - 1:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError(f"rendezvous: only ranks {sorted(served)} of {world - 1} fetched the id")
                srv.settimeout(max(0.1, left))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    try:
                        conn.settimeout(2.0)
                        hello = _recv_exact(conn, len(_MAGIC) + 4)
                        if hello[: len(_MAGIC)] != _MAGIC or len(hello) != len(_MAGIC) + 4:
                            continue  # not one of ours
                        peer = int.from_bytes(hello[len(_MAGIC):], "little")
                        if not 0 < peer < world:
                            continue
                        conn.sendall(_MAGIC + ident)
                        served.add(peer)
                    except OSError:
                        continue  # that rank will come again
        return ident
    last = None
    while time.monotonic() < deadline:
        try:
            with socket.create_connection((addr, port), timeout=2.0) as conn:
                conn.settimeout(5.0)
                conn.sendall(_MAGIC + int(rank).to_bytes(4, "little"))
                data = _recv_exact(conn, len(_MAGIC) + ID_BYTES)
            if data[: len(_MAGIC)] != _MAGIC or len(data) != len(_MAGIC) + ID_BYTES:
                raise RuntimeError(f"rendezvous: ({addr}, {port}) is not served by rank 0 of this job")
            return data[len(_MAGIC):]
        except (ConnectionRefusedError, ConnectionResetError, socket.timeout, TimeoutError) as exc:
            last = exc
            time.sleep(0.05)
        except OSError as exc:
            last = exc
            time.sleep(0.05)
    raise TimeoutError(f"rendezvous: rank {rank} could not reach rank 0 at ({addr}, {port}): {last}")


# ---------------------------------------------------------------------------------------------------------
# the communicator
# ---------------------------------------------------------------------------------------------------------


class Communicator:
    """One rank of an RCCL communicator (``tsim_dist*``), bound to a HIP device."""

    def __init__(self, device: int, ident: bytes, rank: int, world: int):
        if len(ident) != ID_BYTES:
            raise ValueError(f"unique id must be {ID_BYTES} bytes")
        self._lib = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        h = C.c_void_p()
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(ident)
        _lib.check(self._lib.tsim_dist_init(self.device, buf, self.rank, self.world, C.byref(h)), "tsim_dist_init")
        self._h = h

    @classmethod
    def from_env(cls, device: int | None = None, *, port_offset: int = 1) -> "Communicator":
        """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT as a launcher sets them; the id travels over
        ``MASTER_PORT + port_offset`` (``TSIM_DIST_PORT`` overrides the port)."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        port = int(os.environ.get("TSIM_DIST_PORT", int(os.environ.get("MASTER_PORT", "29500")) + port_offset))
        ident = rendezvous_tcp(rank, world, addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), port=port)
        return cls(dev, ident, rank, world)

    def close(self) -> None:
        h, self._h = self._h, None
        if h:
            self._lib.tsim_dist_destroy(h)

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    # -- collectives on device buffers (asynchronous on `stream`; 0 = the communicator's own stream) ---------
    def gather_rows(self, d_send: int, nbytes: int, d_recv: int, *, root: int = 0, stream: int = 0) -> None:
        _lib.check(self._lib.tsim_dist_gather_rows(self._h, C.c_void_p(d_send), int(nbytes), C.c_void_p(d_recv or 0), int(root),
                                                   C.c_void_p(stream) if stream else None), "tsim_dist_gather_rows")

    def alltoall_rows(self, d_send: int, d_recv: int, nbytes_per_peer: int, *, stream: int = 0) -> None:
        _lib.check(self._lib.tsim_dist_alltoall_rows(self._h, C.c_void_p(d_send), C.c_void_p(d_recv), int(nbytes_per_peer),
                                                     C.c_void_p(stream) if stream else None), "tsim_dist_alltoall_rows")

    def stream_wait(self, waiting_stream: int = 0, signalling_stream: int = 0) -> None:
        """``waiting_stream`` waits for what is queued on ``signalling_stream`` (0 = the communicator's stream)."""
        _lib.check(self._lib.tsim_dist_stream_wait(self._h, C.c_void_p(waiting_stream) if waiting_stream else None,
                                                   C.c_void_p(signalling_stream) if signalling_stream else None),
                   "tsim_dist_stream_wait")

    def mark(self, mark: int, stream: int = 0) -> None:
        """Record marker ``mark`` on ``stream`` (e.g. right after a collective was queued there)."""
        _lib.check(self._lib.tsim_dist_mark(self._h, int(mark), C.c_void_p(stream) if stream else None), "tsim_dist_mark")

    def wait_mark(self, mark: int, stream: int = 0) -> None:
        """``stream`` waits for marker ``mark`` only (never recorded: no-op)."""
        _lib.check(self._lib.tsim_dist_wait_mark(self._h, int(mark), C.c_void_p(stream) if stream else None), "tsim_dist_wait_mark")

    # -- blocking host-value helpers ----------------------------------------------------------------------------
    def allreduce_max(self, value: float) -> float:
        v = C.c_double(float(value))
        _lib.check(self._lib.tsim_dist_allreduce_max(self._h, C.byref(v)), "tsim_dist_allreduce_max")
        return float(v.value)

    def barrier(self) -> None:
        _lib.check(self._lib.tsim_dist_barrier(self._h), "tsim_dist_barrier")

    def gather_host(self, hp, local: np.ndarray, *, root: int = 0):
        """Gather equally shaped host arrays through the GPU (H2D, ``ncclGather`` over xGMI, D2H on the root):
        ``[world, *local.shape]`` on ``root``, ``None`` elsewhere.  ``hp``: this rank's :class:`HipProgram`."""
        local = np.ascontiguousarray(local)
        nbytes = local.nbytes
        d_send = hp.malloc(max(1, nbytes))
        d_recv = hp.malloc(max(1, nbytes * self.world)) if self.rank == root else None
        hp.h2d(d_send, local)
        self.gather_rows(d_send.ptr, nbytes, d_recv.ptr if d_recv else 0, root=root)
        self.barrier()  # also drains the communicator's stream, where the gather ran
        out = None
        if self.rank == root:
            out = np.empty((self.world,) + local.shape, dtype=local.dtype)
            hp.d2h(out, d_recv)
            d_recv.free()
        d_send.free()
        return out


# ---------------------------------------------------------------------------------------------------------
# sharding logic (independent of the sampling backend and of the transport)
# ---------------------------------------------------------------------------------------------------------


def hip_sample_fn(device: int = 0, bit_packed: bool = False):
    """The product ``sample_fn``: one shard through the fused kernels on ``device``."""
    from .backend import get_hip_program

    def fn(program, f_rows, key, shot_offset):
        out, _ = get_hip_program(program, device).sample_batch(f_rows, key, shot_offset=shot_offset, bit_packed=bit_packed)
        return out

    return fn


def sample_program_sharded(program, f_params: np.ndarray, key, *, rank: int, world: int, sample_fn, gather, root: int = 0):
    """Every rank passes the same ``(program, f_params, key)``; rank ``root`` returns the full ``[B, ...]`` result,
    the others ``None``.

    ``sample_fn(program, f_rows, key, shot_offset) -> uint8/bool[rows, width]`` samples a shard;
    ``gather(local uint8[rows_max, width]) -> [world, rows_max, width] on root / None elsewhere`` moves the rows
    (:meth:`Communicator.gather_host` in the product).  Shards are padded to the largest one for the gather."""
    f = np.asarray(f_params)
    bounds = shard_bounds(int(f.shape[0]), world)
    lo, hi = bounds[rank]
    local = np.ascontiguousarray(sample_fn(program, f[lo:hi], key, lo)).view(np.uint8)
    width = local.shape[1] if local.ndim == 2 else 0
    rows_max = max(b - a for a, b in bounds)
    padded = np.zeros((rows_max, width), dtype=np.uint8)
    padded[: hi - lo] = local
    parts = gather(padded)
    if rank != root:
        return None
    return np.concatenate([np.asarray(parts[q])[: bounds[q][1] - bounds[q][0]] for q in range(world)], axis=0)


def sample_program_multi_device(program, f_params: np.ndarray, key, devices: list[int]) -> np.ndarray:
    """Single-process variant: shard one batch over several GPUs of this process (one handle and stream per
    device, launches overlap, results concatenated on the host)."""
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
        hp, n, d_o, keep = item
        packed = np.zeros((n, wo * 8), np.uint8)
        hp.d2h(packed, d_o)  # synchronises that device's stream
        parts.append(np.unpackbits(packed, axis=1, bitorder="little")[:, :n_out].view(np.bool_))
        for buf in (d_o, *keep):
            buf.free()
    return np.concatenate(parts, axis=0)


# ---------------------------------------------------------------------------------------------------------------------
# Group arithmetic of the pipelined collection (bench.py --gpus N; tests/test_dist_gloo.py runs it for N = 2, 4, 8 over gloo).
# Every rank writes the bit_packed rows of consecutive batches ("steps") into one of two group buffers; a complete group -
# `gather_every` batches - or the partial one at the end of a region is collected by ONE collective:
#   "root0"    ncclGather to rank 0: rank 0 receives [rank 0's group][rank 1's group] ... (the north star's gather);
#   "alltoall" the same gather with its roots spread: the group is cut into `world` equal chunks of consecutive batches and
#              chunk j of every rank is assembled on rank j: rank j receives [rank 0's chunk j][rank 1's chunk j] ...
#              (needs count % world == 0: a partial group goes to rank 0 instead).
# ---------------------------------------------------------------------------------------------------------------------
XGMI_LINK_BYTES_PER_S = 64e9  # one direction of one point-to-point xGMI link between two GPUs of a node (MI355X_MICROARCH.md)


def gather_group_size(steps: int, world: int, spread_roots: bool, override: int | None = None) -> int:
    """Batches per collective: about a third of a timed region in units of max(4, world), at least 8, at most 64 - every
    collective but the last of a region then hides under the next group's kernels; a multiple of `world` when the roots are spread."""
    if override:
        g = max(1, int(override))
    else:
        unit = max(4, world)
        g = max(8, min(64, (steps // 3) // unit * unit))
    if spread_roots:
        g = (g + world - 1) // world * world
    return g


def group_pieces(step_no: int, k: int, gather_every: int):
    """The next `k` steps, starting at running step number `step_no`, cut at group boundaries: tuples
    (group buffer 0/1, first position inside the group, number of steps)."""
    done = 0
    while done < k:
        j = step_no + done
        g, pos = (j // gather_every) & 1, j % gather_every
        n = min(k - done, gather_every - pos)
        yield g, pos, n
        done += n


def collective_kind(mode: str, count: int, world: int) -> str:
    return "alltoall" if (mode == "alltoall" and count % world == 0) else "root0"


def received_layout(kind: str, count: int, world: int, rank: int) -> list[tuple[int, int, int]]:
    """What rank `rank` holds after the collective of a group of `count` batches, in receive-buffer order:
    (sender, first batch inside the group, number of batches)."""
    if kind == "root0":
        return [(s, 0, count) for s in range(world)] if rank == 0 else []
    per = count // world
    return [(s, rank * per, per) for s in range(world)]


def collection_prediction(one_gpu_shots_per_s: float, row_bytes: int, worlds=(1, 2, 4, 8)) -> dict:
    """Whole-node rate the collection allows, from the one-GPU rate and the link arithmetic (a PREDICTION to read the first
    multi-GPU measurement against): every rank produces `row_bytes` per shot; root0 puts the rows of world - 1 peers on
    rank 0's world - 1 links, alltoall puts 1 / world of a rank's rows on each of its links."""
    out = {}
    for w in worlds:
        if w == 1:
            out["1"] = {"root0": one_gpu_shots_per_s, "alltoall": one_gpu_shots_per_s}
            continue
        per_link_rows = XGMI_LINK_BYTES_PER_S / row_bytes
        out[str(w)] = {"root0": one_gpu_shots_per_s + (w - 1) * min(one_gpu_shots_per_s, per_link_rows),
                       "alltoall": w * min(one_gpu_shots_per_s, per_link_rows * w)}
    return out


class GroupCollector:
    """The double-buffered group collection on the host (numpy buffers, any transport): the model bench.py's device path
    follows, and what the gloo tests drive.  ``transport(kind, send: bytes-like [count * unit], count) -> received bytes or None``."""

    def __init__(self, world: int, rank: int, gather_every: int, mode: str, unit: int, transport):
        self.world, self.rank, self.every, self.mode, self.unit, self.transport = world, rank, gather_every, mode, unit, transport
        self.buf = [np.zeros(gather_every * unit, np.uint8) for _ in range(2)]
        self.step_no = 0
        self.gathered = 0       # groups whose collective has been issued
        self.collected = []     # (first step of the group, kind, count, received array or None)

    def steps(self, payloads) -> None:
        """`payloads`: one ``uint8[unit]`` per step, in step order."""
        it = iter(payloads)
        k = len(payloads)
        for g, pos, n in group_pieces(self.step_no, k, self.every):
            for i in range(n):
                self.buf[g][(pos + i) * self.unit:(pos + i + 1) * self.unit] = next(it)
            self.step_no += n
            if pos + n == self.every:
                self._collect(self.every)

    def drain(self) -> None:
        while self.gathered * self.every < self.step_no:
            self._collect(min(self.every, self.step_no - self.gathered * self.every))
        self.gathered = 0
        self.step_no = 0

    def _collect(self, count: int) -> None:
        g = self.gathered & 1
        kind = collective_kind(self.mode, count, self.world)
        got = self.transport(kind, self.buf[g][: count * self.unit], count)
        self.collected.append((self.gathered * self.every, kind, count, got))
        self.gathered += 1
