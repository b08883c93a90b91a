"""Compiled samplers: the host side of the path above the C ABI.

The classes, keyword surface and RNG consumption are the reference's (``src/tsim/sampler.py:170-953``:
``CompiledMeasurementSampler``, ``CompiledDetectorSampler``, ``CompiledStateProbs``) - for a fixed
``(seed, batch_size)`` and flags the host-noise results are the ones the reference's loop produces:

* one split of the sampler key per batch handed to ``sample_program`` (``sampler.py:399``), one for a
  separately computed reference sample (``:272``), one per dispatched survivor batch (``:482``);
* the channel sampler is seeded with ``default_rng(seed).integers(0, 2**30)`` (``:203``) and asked for
  ``batch_size`` rows per batch (``:393``), or per chunk under post-selection (``:514``);
* a requested reference sample rides as row 0 of the first batch (``:384-404``); survivors of direct
  post-selection are compacted into batches of ``batch_size`` in shot order, the last one padded with its
  first row (``:466-508``).

The orchestration is organised differently: a sampler is built from an already compiled program plus the
noise model (``channel_probs``, ``error_transform``; e.g. ``from_npz``), and there are two *routes*:

* the **device route** (the product): error rows are generated packed, every batch runs through the
  pipelined C-ABI launches into ONE device-resident result buffer, post-selection is the direct-detector
  filter kernel plus row gather/scatter by survivor index, and the request is downloaded once, in the layout
  the caller asked for (bools, or bit-packed bytes compacted on the GPU);
* the **seam route**: when ``tsim_amd.sampler.sample_program`` - a module attribute resolved at call time,
  like the reference's - has been replaced (tests spying on or substituting the backend, exactly as the
  reference's own tests do), batches go through that callable on host arrays in the reference layout.

``noise="device"`` swaps the numpy channel stream for the device-side sampler (``k_noise``): statistically
equivalent, the error bits never leave HBM.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import prng
from .backend import (  # noqa: F401
    alloc_pinned_numpy,
    result_pool,
    DeviceNoiseSampler,
    HipProgram,
    check_norm_deviation,
    evaluate,
    get_hip_program,
    sample_program,
)
from .channels import ChannelSampler
from .program import CompiledProgram, from_tsim, load_npz

_backend_sample_program = sample_program

_MAX_AUTO_BATCH = 1 << 24  # shots per launch beyond which nothing is gained
_LANES = 8                 # launches in flight on the device route (one f buffer each)


# ---------------------------------------------------------------------------------------------------------
# batch arithmetic
# ---------------------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class BatchPlan:
    """``count`` equal batches of ``size`` rows."""

    size: int
    count: int


def plan_batches(shots: int, batch_size: int | None, cap: int, *, reserve_row: bool = False) -> BatchPlan:
    """Equal batches that cover ``shots``: ``batch_size`` as given, else the fewest batches of at most ``cap``
    rows, evenly filled.  ``reserve_row``: one row of the first batch is spoken for (the noiseless reference
    row), so a plan that covers ``shots`` exactly grows by one row per batch - shapes stay uniform
    (the table of test/unit/test_sampler.py:288-346)."""
    ceil_div = lambda a, b: -(-a // b)  # noqa: E731
    count = ceil_div(shots, batch_size) if batch_size is not None else max(1, ceil_div(shots, cap))
    size = batch_size if batch_size is not None else ceil_div(shots, count)
    spare = size * count - shots  # rows sampled beyond the request
    return BatchPlan(size + (1 if reserve_row and spare == 0 else 0), count)


def _check_request(shots: int, batch_size: int | None) -> None:
    if shots < 0:
        raise ValueError(f"shots must be non-negative, got {shots}")
    if batch_size is not None and batch_size < 1:
        raise ValueError(f"batch_size must be at least 1, got {batch_size}")


# ---------------------------------------------------------------------------------------------------------
# outputs that are one f bit (xor a constant): no amplitude needed
# ---------------------------------------------------------------------------------------------------------


class _DirectOutputs:
    def __init__(self, program: CompiledProgram):
        self.f_index = np.asarray(program.direct_f_indices, dtype=np.int64)
        self.flip = np.asarray(program.direct_flips, dtype=np.bool_)
        self.column = np.asarray(program.output_order[: len(self.f_index)], dtype=np.int64)
        self.num_outputs = int(program.num_outputs)
        is_direct = np.zeros(self.num_outputs, dtype=np.bool_)
        is_direct[self.column] = True
        self.detector_mask = is_direct[: int(program.num_detectors)].copy()

    def fill(self, f_bits: np.ndarray) -> np.ndarray:
        """``bool[n, num_outputs]``: the direct columns from ``f_bits`` (uint8 ``[n, num_f]``), False elsewhere."""
        out = np.zeros((f_bits.shape[0], self.num_outputs), dtype=np.bool_)
        if len(self.f_index):
            out[:, self.column] = f_bits[:, self.f_index].astype(np.bool_) ^ self.flip
        return out


# ---------------------------------------------------------------------------------------------------------
# post-selection on directly readable detectors: one driver, two ways of doing the work
# ---------------------------------------------------------------------------------------------------------


def _run_postselected(shots: int, size: int, work, next_key) -> None:
    """Shots arrive in chunks of ``size``; ``work.admit`` reports the survivors of a chunk; survivors queue
    up in shot order and leave in full batches of ``size`` (``work.dispatch``), the last batch partially
    filled.  One key per dispatched batch, drawn when it is dispatched."""
    queue = np.zeros(0, dtype=np.int64)
    done = 0
    while done < shots:
        n = min(size, shots - done)
        queue = np.concatenate([queue, done + work.admit(done, n)])
        done += n
        while len(queue) >= size:
            work.dispatch(queue[:size], size, next_key())
            queue = queue[size:]
    if len(queue):
        work.dispatch(queue, size, next_key())


def _run_postselected_device(shots: int, size: int, work, next_key, sample_packed) -> None:
    """The same schedule with the survivors queued ON THE DEVICE (``_DevicePostselect.admit_device``): per chunk only the
    queue length crosses PCIe; the channel sampler runs a chunk ahead on a worker thread (stream order is chunk order)."""
    import queue
    import threading

    chunks = [min(size, shots - lo) for lo in range(0, shots, size)]
    ready: "queue.Queue" = queue.Queue(maxsize=2)

    def produce() -> None:
        try:
            for n in chunks:
                ready.put(sample_packed(n))
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            ready.put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    try:
        done = head = tail = 0
        for n in chunks:
            rows = ready.get()
            if isinstance(rows, BaseException):
                raise rows
            tail = work.admit_device(done, rows)
            done += n
            while tail - head >= size:
                work.dispatch_device(head, size, size, next_key())
                head += size
        if tail > head:
            work.dispatch_device(head, tail - head, size, next_key())
    finally:
        while th.is_alive():  # (an error above: let the producer run out so that the generator is not left mid-draw)
            try:
                ready.get(timeout=0.05)
            except queue.Empty:
                pass
        th.join()


class _SeamPostselect:
    """Post-selection work through the ``sample_program`` seam, host arrays in the reference layout."""

    def __init__(self, owner, shots: int, test_mask: np.ndarray, ref_det: np.ndarray | None):
        self.owner, self.mask, self.ref = owner, test_mask, ref_det
        self.nd = owner._num_detectors
        self.f = np.zeros((shots, owner._channel_sampler.num_f), dtype=np.uint8)
        self.rows = np.zeros((shots, owner._program.num_outputs), dtype=np.bool_)
        self.discarded = np.zeros(shots, dtype=np.bool_)

    def admit(self, start: int, n: int) -> np.ndarray:
        f = self.owner._channel_sampler.sample(n)
        self.f[start:start + n] = f
        det = self.owner._direct.fill(f)[:, : self.nd]
        fired = (det if self.ref is None else det ^ self.ref) & self.mask
        gone = fired.any(axis=1)
        self.rows[start:start + n, : self.nd] = det
        self.discarded[start:start + n] = gone
        return np.flatnonzero(~gone)

    def dispatch(self, shot_ids: np.ndarray, size: int, key) -> None:
        f = self.f[shot_ids]
        if len(shot_ids) < size:  # fixed batch shape: fill up with the first survivor
            f = np.concatenate([f, np.repeat(f[:1], size - len(shot_ids), axis=0)])
        out = np.asarray(self.owner._seam(f, key))
        self.rows[shot_ids] = out[: len(shot_ids)]

    def collect(self):
        return self.rows, self.discarded


class _DevicePostselect:
    """The same work in HBM: every chunk's packed rows are appended to a device-resident store, the
    direct-detector filter kernel writes the direct bits of every row and flags the discarded ones, survivor
    batches are gathered by index, sampled densely (Threefry counter = position in the batch, as in the
    reference's compacted batches) and their result rows scattered back to the shots they belong to."""

    def __init__(self, owner, shots: int, size: int, test_mask: np.ndarray, ref_det: np.ndarray | None):
        self.owner = owner
        hp = self.hp = owner._hip()
        prog = owner._program
        self.num_f = owner._channel_sampler.num_f
        self.n_out, self.nd = int(prog.num_outputs), owner._num_detectors
        self.wf, self.wo = max(1, (self.num_f + 63) // 64), (self.n_out + 63) // 64
        self.shots, self.size = shots, size
        self.n_comp = max(1, len(prog.components))
        self.devs = []
        # the store, the rows, the flags and the unpack scratch are O(shots) on the device: refuse what cannot fit instead
        # of failing half way
        need = shots * (self.wf * 8 + self.wo * 8 + 1 + 4 + self.n_out) + size * (self.wf * 8 + self.wo * 8 + 8)
        free, _ = hp.mem_info()
        if need > 0.9 * free:
            raise MemoryError(f"device post-selection of {shots} shots needs {need >> 20} MiB of device memory, {free >> 20} MiB are free: "
                              "sample in several calls")
        self._bufs = []
        try:
            self._allocate(hp, shots, size, test_mask, ref_det)
        except BaseException:  # a malloc / upload failed half way: give back what was taken (the caller never sees `work`)
            self.release()
            raise

    def _allocate(self, hp, shots: int, size: int, test_mask: np.ndarray, ref_det) -> None:
        def m(nbytes):
            self._bufs.append(hp.malloc(nbytes))
            return self._bufs[-1]

        self.d_store, self.d_rows = m(shots * self.wf * 8), m(shots * self.wo * 8)
        self.d_flags, self.d_list, self.d_count = m(shots), m(size * 4), m(4)
        self.d_fb, self.d_ob, self.d_idx = m(size * self.wf * 8), m(size * self.wo * 8), m(size * 4)
        self.d_dev, self.d_mask, self.d_ref = m(4 * self.n_comp), m(self.wo * 8), m(self.wo * 8)
        # the survivor queue (shot ids in shot order, appended chunk by chunk on the device), its tail, scan scratch, and one
        # normalisation-deviation record per dispatched batch (read once, at the end)
        self.max_dispatch = shots // max(1, size) + 2
        self.d_queue, self.d_tail, self.d_scan = m(shots * 4 + 16), m(16), m(((size + 1023) // 1024) * 4 + 16)
        self.d_devs = m(4 * self.n_comp * self.max_dispatch)
        self.n_dispatch = 0
        hp.h2d(self.d_tail, np.zeros(4, dtype=np.uint32))
        hp.h2d(self.d_devs, np.zeros(self.n_comp * self.max_dispatch, dtype=np.float32))
        hp.h2d(self.d_mask, self._columns(test_mask))
        self.has_ref = ref_det is not None
        if self.has_ref:
            hp.h2d(self.d_ref, self._columns(ref_det))

    def _columns(self, det_bits: np.ndarray) -> np.ndarray:
        """Detector bits -> packed output-row words (detectors are the first columns)."""
        full = np.zeros(self.wo * 64, dtype=np.uint8)
        full[: self.nd] = np.asarray(det_bits, dtype=np.uint8)
        return np.packbits(full, bitorder="little").view(np.uint64)

    def admit(self, start: int, n: int) -> np.ndarray:
        hp = self.hp
        rows = self.owner._channel_sampler.sample_packed(n)
        hp.h2d(self.d_store.ptr + start * self.wf * 8, rows)
        hp.postselect_device(self.d_store.ptr + start * self.wf * 8, n, self.num_f, self.d_mask.ptr,
                             self.d_ref.ptr if self.has_ref else 0, self.d_rows.ptr + start * self.wo * 8,
                             self.d_list.ptr, self.d_count.ptr, self.d_flags.ptr + start)
        gone = np.empty(n, dtype=np.uint8)
        hp.d2h(gone, self.d_flags.ptr + start)
        return np.flatnonzero(gone == 0)

    def dispatch(self, shot_ids: np.ndarray, size: int, key) -> None:
        hp, n = self.hp, len(shot_ids)
        hp.h2d(self.d_idx, shot_ids.astype(np.uint32))
        hp.gather_rows_device(self.d_store.ptr, self.wf, self.d_idx.ptr, n, size, self.d_fb.ptr)
        hp.sample_batch_device(self.d_fb.ptr, size, self.num_f, key, self.d_ob.ptr, d_norm_dev=self.d_dev.ptr)
        hp.scatter_rows_device(self.d_ob.ptr, self.wo, self.d_idx.ptr, n, self.d_rows.ptr)
        dev = np.zeros(self.n_comp, dtype=np.float32)
        hp.d2h(dev, self.d_dev)
        self.devs.append(dev)

    def admit_device(self, start: int, rows: np.ndarray) -> int:
        """A chunk's packed f rows -> store, direct bits + discard flags, survivors appended (in shot order) to the device
        queue; returns the queue's length - the only thing that comes back (4 bytes)."""
        hp, n = self.hp, len(rows)
        hp.h2d(self.d_store.ptr + start * self.wf * 8, rows)
        hp.postselect_device(self.d_store.ptr + start * self.wf * 8, n, self.num_f, self.d_mask.ptr,
                             self.d_ref.ptr if self.has_ref else 0, self.d_rows.ptr + start * self.wo * 8,
                             self.d_list.ptr, self.d_count.ptr, self.d_flags.ptr + start)
        hp.survivors_append_device(self.d_flags.ptr + start, n, start, self.d_scan.ptr, self.d_queue.ptr, self.d_tail.ptr)
        tail = np.zeros(1, dtype=np.uint32)
        hp.d2h(tail, self.d_tail.ptr)
        return int(tail[0])

    def dispatch_device(self, head: int, n_valid: int, size: int, key) -> None:
        """One dense batch of survivors queue[head : head + n_valid] (padded to ``size`` with its first row, as the
        reference pads, sampler.py:489-496): gather, sample, scatter - no host data."""
        hp = self.hp
        idx = self.d_queue.ptr + head * 4
        hp.gather_rows_device(self.d_store.ptr, self.wf, idx, n_valid, size, self.d_fb.ptr)
        hp.sample_batch_device(self.d_fb.ptr, size, self.num_f, key, self.d_ob.ptr,
                               d_norm_dev=self.d_devs.ptr + self.n_dispatch * self.n_comp * 4)
        hp.scatter_rows_device(self.d_ob.ptr, self.wo, idx, n_valid, self.d_rows.ptr)
        self.n_dispatch += 1

    def release(self) -> None:
        """Give the device buffers back - also on the error paths (a normalisation error, a failed allocation or kernel)."""
        for buf in self._bufs:
            buf.free()
        self._bufs = []

    def collect_device(self, post: dict, packed_columns: int | None):
        """Blanking and reference bits on the device (tsim_postselect_rows_device: the filter's own test on the finished
        rows), then either bools or - ``packed_columns`` - the reference's bit_packed rows of that many leading columns:
        no fancy indexing over an 80 MB bool array, no host packbits."""
        hp, owner = self.hp, self.owner
        row_bytes = self.wo * 8

        def as_row(cols) -> np.ndarray:
            full = np.zeros(row_bytes * 8, dtype=np.uint8)
            full[: self.n_out] = np.asarray(cols, dtype=np.uint8)[: self.n_out]
            return np.packbits(full, bitorder="little")

        masks = np.concatenate([as_row(post[k]) for k in ("test", "test_ref", "keep", "xor_kept", "xor_discarded")])
        d_masks = hp.malloc(masks.nbytes + 16)
        self._bufs.append(d_masks)
        hp.h2d(d_masks, masks)
        hp.postselect_rows_device(self.d_rows.ptr, self.shots, row_bytes, d_masks.ptr, self.d_flags.ptr)
        gone = np.empty(self.shots, dtype=np.uint8)
        hp.d2h(gone, self.d_flags)
        if packed_columns is not None:
            rb = (packed_columns + 7) // 8
            d_c = hp.malloc(self.shots * rb + 16)
            self._bufs.append(d_c)
            hp.compact_rows_device(self.d_rows.ptr, self.shots, packed_columns, d_c.ptr, in_words=self.wo)
            rows = np.empty((self.shots, rb), dtype=np.uint8)
            hp.d2h(rows, d_c)
        else:
            rows = owner._download_bools(hp, self.d_rows, self.shots)
        if self.n_dispatch:
            devs = np.zeros(self.n_dispatch * self.n_comp, dtype=np.float32)
            hp.d2h(devs, self.d_devs.ptr)
            self.devs += [devs[k * self.n_comp:(k + 1) * self.n_comp] for k in range(self.n_dispatch)]
        self.release()
        for dev in self.devs:
            owner._check_devs(dev)
        return rows, gone.view(np.bool_)

    def collect(self):
        rows, gone = self.owner._download_bools(self.hp, self.d_rows, self.shots), np.empty(self.shots, dtype=np.uint8)
        self.hp.d2h(gone, self.d_flags)
        self.release()
        for dev in self.devs:
            self.owner._check_devs(dev)
        gone = gone.astype(np.bool_)
        rows[gone, self.nd:] = False  # a discarded shot keeps its direct DETECTOR columns only
        return rows, gone


# ---------------------------------------------------------------------------------------------------------
# the samplers
# ---------------------------------------------------------------------------------------------------------


class _CompiledSamplerBase:
    def __init__(self, program, *, channel_probs: list, error_transform: np.ndarray, seed: int | None = None,
                 device: int = 0, noise: str = "host", mode: str = "auto"):
        """``noise="host"``: the reference's numpy channel stream, bit for bit; ``noise="device"``: channels
        sampled on the GPU.  ``mode``: kernel formulation, see :class:`tsim_amd.backend.HipProgram`.

        ``batch_size`` in ``sample()``: with ``noise="host"`` it is the reference's batch size (the channel stream and the
        key chain depend on it, ``sampler.py:377-399``).  With ``noise="device"`` there is no stream to reproduce, and a
        ``batch_size`` below 2^20 is REPLACED by batches of about 2^20 shots (results for a fixed seed are those of that
        batching, whatever ``batch_size`` was); the result buffers on the device are O(shots) either way, so on this path
        ``batch_size`` does not bound device memory - split the request into several ``sample()`` calls for that."""
        if noise not in ("host", "device"):
            raise ValueError("noise must be 'host' or 'device'")
        if seed is None:
            seed = int(np.random.default_rng().integers(0, 2**30))
        self._noise, self._mode, self._device = noise, mode, int(device)
        self._key = prng.key(seed)
        self._program: CompiledProgram = from_tsim(program)
        channel_seed = int(np.random.default_rng(seed).integers(0, 2**30))
        self._channel_sampler = ChannelSampler(channel_probs=channel_probs, error_transform=error_transform, seed=channel_seed)
        self._noise_key = prng.key(channel_seed)  # key chain of the device-side channel sampler
        self._num_detectors = int(self._program.num_detectors)
        self._direct = _DirectOutputs(self._program)
        self._direct_detector_mask = self._direct.detector_mask
        self._device_noise = None
        self._f_slots = None
        self._f_ring = None
        self._stage = None      # host staging array of one batch of packed f rows
        self._bufs: dict = {}  # name -> (HipProgram, DeviceBuffer): grow-only scratch of the device route

    @classmethod
    def from_npz(cls, path, *, seed: int | None = None, device: int = 0, **kw):
        """A program + noise model written by ``program.save_npz(path, prog, n_channels=, channel_probs_<i>=,
        error_transform=)`` where tsim is installed."""
        prog, extra = load_npz(path)
        probs = [np.asarray(extra[f"channel_probs_{i}"], dtype=np.float64) for i in range(int(extra["n_channels"]))]
        return cls(prog, channel_probs=probs, error_transform=extra["error_transform"], seed=seed, device=device, **kw)

    # -- routes ---------------------------------------------------------------------------------------------
    def _seam_replaced(self) -> bool:
        return globals()["sample_program"] is not _backend_sample_program

    def _seam(self, f_params: np.ndarray, key) -> np.ndarray:
        """Call whatever is bound to this module's ``sample_program`` right now."""
        fn = globals()["sample_program"]
        if fn is _backend_sample_program:
            return fn(self._program, f_params, key, device=self._device, mode=self._mode)
        return fn(self._program, f_params, key)

    def _hip(self) -> HipProgram:
        return get_hip_program(self._program, self._device, self._mode)

    def _next_key(self):
        self._key, sub = prng.split(self._key)
        return sub

    def _check_devs(self, devs) -> None:
        for dev in np.asarray(devs, dtype=np.float32).reshape(-1)[: len(self._program.components)]:
            check_norm_deviation(float(dev))

    # -- sizing ----------------------------------------------------------------------------------------------
    def _bytes_per_shot(self) -> int:
        """HBM per shot on the device route: packed f row + packed result row + its unpacked bytes.  (The
        reference's estimate counts its materialised [B, G, T, 4] tensors, sampler.py:294-306; here those
        live in registers.)"""
        num_f, n_out = self._channel_sampler.num_f, int(self._program.num_outputs)
        return max(1, 8 * ((num_f + 63) // 64) + 8 * ((n_out + 63) // 64) + n_out)

    def _estimate_batch_size(self) -> int:
        """Largest batch worth launching: half of the device's free memory (queried through the C ABI, as
        sampler.py:308-320 asks jax), at most 2**24 shots."""
        if self._seam_replaced() or not self._program.components:
            return 1 << 20
        free, _total = self._hip().mem_info()
        return max(1, min(_MAX_AUTO_BATCH, (free // 2) // self._bytes_per_shot()))

    # -- the noiseless sample ------------------------------------------------------------------------------------
    def _compute_reference_sample(self) -> np.ndarray:
        """The outputs of the noiseless circuit (all f = 0); costs a key split iff something is compiled
        (sampler.py:263-276)."""
        quiet = np.zeros((1, self._channel_sampler.num_f), dtype=np.uint8)
        rows = self._seam(quiet, self._next_key()) if self._program.components else self._direct.fill(quiet)
        return np.asarray(rows[0], dtype=np.bool_)

    def _compute_direct_outputs(self, f_params_np: np.ndarray) -> np.ndarray:
        return self._direct.fill(np.asarray(f_params_np))

    def _sample_direct(self, shots: int) -> np.ndarray:
        """Programs without compiled components: pure host work, the device is never touched
        (sampler.py:547-555)."""
        # fill() places every direct bit at its final column (output_order), i.e. the result is already reindexed
        return self._direct.fill(self._channel_sampler.sample(shots))

    # -- plain batching ----------------------------------------------------------------------------------------
    def _sample_batches(self, shots: int, batch_size: int | None = None, *, compute_reference: bool = False):
        _check_request(shots, batch_size)
        n_out = int(self._program.num_outputs)
        if shots == 0:
            rows, ref = np.empty((0, n_out), dtype=np.bool_), np.zeros(n_out, dtype=np.bool_)
        elif not self._program.components:
            rows = self._direct_device(shots, batch_size) if self._direct_on_device(shots) else self._sample_direct(shots)
            ref = self._compute_reference_sample() if compute_reference else None
        elif self._seam_replaced():
            plan = plan_batches(shots, batch_size, self._estimate_batch_size(), reserve_row=compute_reference)
            rows, ref = self._seam_plain(shots, plan, compute_reference)
        else:
            rows, ref = self._device_plain(shots, batch_size, compute_reference)
        return (rows, ref) if compute_reference else rows

    # -- programs without compiled components (Clifford-only circuits) on the device ----------------------------------
    def _direct_on_device(self, shots: int) -> bool:
        """``noise="device"``: always (that is what was asked for).  ``noise="host"``: when a GPU is there and the request is
        large enough to pay for the trip - the reference does this on the host (``_sample_direct``, sampler.py:547-555), the
        numbers are the same either way (one ``ChannelSampler`` call for all shots, a deterministic function of its rows)."""
        if self._seam_replaced() or int(self._program.num_outputs) == 0:
            return False
        if self._noise == "device":
            return True
        if shots < (1 << 15) or self._channel_sampler._native is None:
            return False
        try:
            from . import _lib

            _lib.load(build=False)
            return _lib.device_count() > 0
        except Exception:  # noqa: BLE001 - no library / no device: the host path is complete on its own
            return False

    def _direct_device(self, shots: int, batch_size: int | None, packed_columns: int | None = None) -> np.ndarray:
        """Direct outputs by the streaming kernel (``tsim_direct.hip.h`` through ``tsim_sample_steps_device``).  Device noise:
        the three-stage pipeline of :meth:`_device_noise_plain`.  Host noise: ONE ``sample_packed(shots)`` call - the
        reference's stream for this request (sampler.py:549) - uploaded and processed in chunks."""
        if self._noise == "device":
            return self._device_noise_plain(shots, batch_size, False, packed_columns)[0]
        hp = self._hip()
        cs = self._channel_sampler
        num_f, n_out = cs.num_f, int(self._program.num_outputs)
        wf, wo = max(1, (num_f + 63) // 64), (n_out + 63) // 64
        f_all = cs.sample_packed(shots)
        chunk = min(shots, 1 << 20)
        direct_packed = packed_columns is not None and packed_columns == n_out
        row_bytes = (n_out + 7) // 8 if direct_packed else wo * 8
        d_f, d_rows = self._scratch(hp, "direct_f", chunk * wf * 8), self._scratch(hp, "rows", chunk * row_bytes + 16)
        if packed_columns is not None:
            rb = (packed_columns + 7) // 8
            out = np.empty((shots, rb), dtype=np.uint8)
            d_c = d_rows if direct_packed else self._scratch(hp, "compact", chunk * rb + 16)
        else:
            out = np.empty((shots, n_out), dtype=np.uint8)
            d_u8 = self._scratch(hp, "unpacked", chunk * n_out)
        key_state = (C.c_uint32 * 2)(0, 0)  # (no random numbers in a program without components)
        for lo in range(0, shots, chunk):
            n = min(chunk, shots - lo)
            hp.h2d(d_f, f_all[lo:lo + n])
            hp.sample_steps_device([d_f.ptr], n, num_f, key_state, [d_rows.ptr], out_bit_packed=direct_packed)
            hp.synchronize()
            if packed_columns is not None:
                if not direct_packed:
                    hp.compact_rows_device(d_rows.ptr, n, packed_columns, d_c.ptr, in_words=wo)
                hp.d2h(out[lo:lo + n], d_c.ptr)
            else:
                hp.unpack_bits_device(d_rows.ptr, n, n_out, d_u8.ptr)
                hp.d2h(out[lo:lo + n], d_u8.ptr)
        return out if packed_columns is not None else out.view(np.bool_)

    def _seam_plain(self, shots: int, plan: BatchPlan, want_ref: bool):
        parts, ref = [], None
        for b in range(plan.count):
            f = self._channel_sampler.sample(plan.size)
            rides = want_ref and b == 0
            if rides:
                f[0] = 0
            out = np.asarray(self._seam(f, self._next_key()))
            if rides:
                ref, out = np.array(out[0], dtype=np.bool_), out[1:]
            parts.append(out)
        rows = parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)
        return np.asarray(rows[:shots], dtype=np.bool_), ref

    def _device_noise_sampler(self, hp: HipProgram) -> DeviceNoiseSampler:
        if self._device_noise is None or self._device_noise._prog is not hp:
            self._device_noise = DeviceNoiseSampler(hp, self._channel_sampler)
        return self._device_noise

    def _lane_buffers(self, hp: HipProgram, nbytes: int) -> list:
        have = self._f_slots
        if have is None or have[0] is not hp or have[1] < nbytes:
            if have is not None and have[0] is hp:
                hp.synchronize()
                for buf in have[2]:
                    buf.free()
            self._f_slots = have = (hp, nbytes, [hp.malloc(nbytes) for _ in range(_LANES)])
        return have[2]

    def _scratch(self, hp: HipProgram, name: str, nbytes: int):
        """A device buffer of at least ``nbytes`` kept between calls (allocation and free cost more than the
        kernels of a small request); grown when too small, dropped by :meth:`release`."""
        have = self._bufs.get(name)
        if have is None or have[0] is not hp or have[1].nbytes < nbytes:
            if have is not None and have[0] is hp:
                have[1].free()
            have = self._bufs[name] = (hp, hp.malloc(max(1, nbytes)))
        return have[1]

    def release(self) -> None:
        """Give the cached device buffers back (they are otherwise freed with the program's handle)."""
        for _hp, buf in self._bufs.values():
            buf.free()
        self._bufs.clear()
        if self._f_ring is not None:
            for buf in self._f_ring[2]:
                buf.free()
            self._f_ring = None
        if self._f_slots is not None:
            for buf in self._f_slots[2]:
                buf.free()
            self._f_slots = None

    def _download_bools(self, hp: HipProgram, d_rows, n: int) -> np.ndarray:
        """Packed device rows -> ``bool[n, num_outputs]`` on the host (unpack kernel + one D2H into a pageable
        array: a fresh pinned allocation of this size costs more than the copy)."""
        n_out = int(self._program.num_outputs)
        d_u8 = self._scratch(hp, "unpacked", n * n_out)
        hp.unpack_bits_device(d_rows.ptr if hasattr(d_rows, "ptr") else d_rows, n, n_out, d_u8.ptr)
        host = np.empty((n, n_out), dtype=np.uint8)
        hp.d2h(host, d_u8.ptr)
        return host.view(np.bool_)

    # -- output arrangement on the device (sample()'s epilogue, sampler.py:850-868) -------------------------------------
    def _layout_blocks(self, hp: HipProgram, layout: list, total: int) -> list:
        """Per wanted output block: device column table, device staging and a pinned host array."""
        pool = result_pool()
        blocks = []
        for k, (cols, packed) in enumerate(layout):
            cols = np.ascontiguousarray(cols, dtype=np.uint32)
            n_cols = len(cols)
            nbytes = (n_cols + 7) // 8 if packed else n_cols
            blk = {"n_cols": n_cols, "packed": bool(packed), "nbytes": nbytes, "out": pool.take((total, nbytes)) if nbytes else np.empty((total, 0), np.uint8)}
            if nbytes:
                blk["d_cols"] = self._scratch(hp, f"cols{k}", n_cols * 4 + 16)
                hp.h2d(blk["d_cols"], cols)
                blk["d_arr"] = self._scratch(hp, f"arranged{k}", total * nbytes + 16)
            blocks.append(blk)
        return blocks

    def _layout_download(self, hp: HipProgram, blocks: list, d_rows_ptr: int, wo: int, r0: int, r1: int, stream: int) -> None:
        for blk in blocks:
            if not blk["nbytes"]:
                continue
            dst = blk["d_arr"].ptr + r0 * blk["nbytes"]
            hp.arrange_rows_device(d_rows_ptr, r1 - r0, wo, blk["d_cols"].ptr, blk["n_cols"], blk["packed"], dst, stream=stream)
            hp.d2h_async(blk["out"][r0:r1], dst, stream)

    @staticmethod
    def _layout_result(blocks: list, lo: int, hi: int) -> list:
        return [(b["out"][lo:hi] if b["packed"] else b["out"][lo:hi].view(np.bool_)) for b in blocks]

    def _device_noise_plain(self, shots: int, batch_size: int | None, want_ref: bool, packed_columns: int | None = None,
                            post: dict | None = None, layout: list | None = None):
        """``noise="device"``: three things at once - ``k_noise`` fills the f rows of group g + 1 on its own stream, the
        sampling kernels of group g run on the pipeline lanes (``tsim_sample_steps_device``: fused first passes, one
        hard-row batch per group), the rows of group g - 1 travel to the host on a copy stream.  A group is a few
        batches; every batch owns the f buffer of its pipeline slot, so the noise stream only waits for the batch that
        used the slot before (``tsim_sample_batch_device_end`` on it).  The reference's loop does the same things one
        after the other (``sampler.py:393-415``: sample f, upload, sample_program, concatenate, download once)."""
        hp = self._hip()
        # device-side noise has no stream contract with the reference: batch_size only bounds memory here, so batches of
        # fewer than 2^20 shots are merged (batch_size = 10^4: 2.5e8 -> 5e9 shots/s; a fixed seed still gives fixed
        # results) - the host-noise path must honour batch_size, the reference's channel stream depends on it
        if batch_size is not None and batch_size < (1 << 20) and shots > batch_size:
            batch_size = min(1 << 20, self._estimate_batch_size())
        # no batch_size: batches of about 2^20 shots rather than one of everything - noise, sampling and downloads overlap
        # batch against batch (4e6 shots: 4.8e9 -> 8e9 shots/s)
        plan = plan_batches(shots, batch_size, min(1 << 20, self._estimate_batch_size()) if batch_size is None else self._estimate_batch_size())
        ref = self._compute_reference_sample() if want_ref else None
        cs = self._channel_sampler
        num_f, n_out = cs.num_f, int(self._program.num_outputs)
        wf, wo = max(1, (num_f + 63) // 64), (n_out + 63) // 64
        n_comp = max(1, len(self._program.components))
        total, size = plan.size * plan.count, plan.size
        nslot = HipProgram.PIPELINE_SLOTS
        ring = self._f_ring
        if ring is None or ring[0] is not hp or ring[1] < size * wf * 8:
            if ring is not None and ring[0] is hp:
                hp.synchronize()
                for buf in ring[2]:
                    buf.free()
            self._f_ring = ring = (hp, size * wf * 8, [hp.malloc(size * wf * 8) for _ in range(nslot)])
        f_ring = ring[2]
        direct_packed = layout is None and packed_columns is not None and packed_columns == n_out
        row_bytes = (n_out + 7) // 8 if direct_packed else wo * 8
        d_rows, d_devs = self._scratch(hp, "rows", total * row_bytes + 16), self._scratch(hp, "devs", plan.count * n_comp * 4)
        noise = self._device_noise_sampler(hp)
        s_copy = hp.aux_stream(1)
        # results land in recycled pinned memory (backend.PinnedPool): truly asynchronous copies, no first-touch page faults
        pool = result_pool()
        blocks = None
        if layout is not None:  # column selection / order / flips / packing per output block, on the device
            blocks = self._layout_blocks(hp, layout, total)
            out = None
        elif direct_packed:
            out = pool.take((total, row_bytes))
        elif packed_columns is not None:
            rb = (packed_columns + 7) // 8
            out = pool.take((total, rb))
            d_c = self._scratch(hp, "compact", total * rb + 16)
        else:
            out = pool.take((total, n_out))
            d_u8 = self._scratch(hp, "unpacked", total * n_out)
        d_masks = d_gone = None
        if post is not None:
            # post-selection after sampling (tsim_postselect_rows_device): five column masks in the layout of the device rows
            def as_row(cols) -> np.ndarray:
                full = np.zeros(row_bytes * 8, dtype=np.uint8)
                full[:n_out] = np.asarray(cols, dtype=np.uint8)[:n_out]
                return np.packbits(full, bitorder="little")

            masks = np.concatenate([as_row(post[k]) for k in ("test", "test_ref", "keep", "xor_kept", "xor_discarded")])
            d_masks, d_gone = self._scratch(hp, "ps_masks", masks.nbytes + 16), self._scratch(hp, "ps_gone", total + 16)
            hp.h2d(d_masks, masks)
        key_state = (C.c_uint32 * 2)(self._key[0] & 0xFFFFFFFF, self._key[1] & 0xFFFFFFFF)
        # groups: small at both ends (the first one starts the GPU early, the last one is all that is left to download
        # when the kernels are done), up to 4 batches in between
        sizes, left = [], plan.count
        while left > 0:
            n = 1 if (not sizes or left <= 2) else min(4, left - 1)
            sizes.append(n)
            left -= n
        groups, b0, done = [], 0, 0

        def download(g) -> None:
            lo, n, slots = g
            for sl in slots:
                hp.sample_batch_device_end(sl, s_copy)  # the copy stream waits for exactly these batches
            r0, r1 = lo * size, (lo + n) * size
            if post is not None:  # blank the discarded rows where they are, before any layout conversion
                hp.postselect_rows_device(d_rows.ptr + r0 * row_bytes, r1 - r0, row_bytes, d_masks.ptr, d_gone.ptr + r0, stream=s_copy)
            if blocks is not None:
                self._layout_download(hp, blocks, d_rows.ptr + r0 * row_bytes, wo, r0, r1, s_copy)
            elif direct_packed:
                hp.d2h_async(out[r0:r1], d_rows.ptr + r0 * row_bytes, s_copy)
            elif packed_columns is not None:
                hp.compact_rows_device(d_rows.ptr + r0 * row_bytes, r1 - r0, packed_columns, d_c.ptr + r0 * rb, in_words=wo, stream=s_copy)
                hp.d2h_async(out[r0:r1], d_c.ptr + r0 * rb, s_copy)
            else:
                hp.unpack_bits_device(d_rows.ptr + r0 * row_bytes, r1 - r0, n_out, d_u8.ptr + r0 * n_out, stream=s_copy)
                hp.d2h_async(out[r0:r1], d_u8.ptr + r0 * n_out, s_copy)

        noise_key_state = (C.c_uint32 * 2)(self._noise_key[0] & 0xFFFFFFFF, self._noise_key[1] & 0xFFFFFFFF)
        for n in sizes:
            first = hp.pipeline_next_slot()
            slots = [(first + i) % nslot for i in range(n)]
            # noise and sampling of the group in ONE call (round 6, tsim_sample_steps_noise_device): batch j's noise key is the next split
            # of the sampler's noise key chain, its f rows go to the buffer of its pipeline slot - which the library orders behind
            # the slot's previous launch, hard rows included, before anything overwrites it.  One-component programs over narrow
            # rows (the distillation shapes) draw the noise inside their first pass: one kernel per group instead of n + 1.
            hp.sample_steps_noise_device(noise, [f_ring[sl].ptr for sl in slots], size, num_f, key_state, noise_key_state,
                                         [d_rows.ptr + (b0 + i) * size * row_bytes for i in range(n)],
                                         out_bit_packed=direct_packed, d_norm_dev=[d_devs.ptr + (b0 + i) * n_comp * 4 for i in range(n)])
            groups.append((b0, n, slots))
            b0 += n
            if len(groups) >= 2:  # the previous group's rows leave while this group is sampled (pinned target: the call returns at once)
                download(groups[done])
                done += 1
        while done < len(groups):
            download(groups[done])
            done += 1
        self._key = (int(key_state[0]), int(key_state[1]))
        self._noise_key = (int(noise_key_state[0]), int(noise_key_state[1]))
        # the copy stream is behind every batch (download() joined each slot on it): once it has drained, every kernel
        # that writes d_devs has finished - only then may the handle's stream read them
        hp.stream_synchronize(s_copy)
        devs = np.zeros(plan.count * n_comp, dtype=np.float32)
        hp.d2h(devs, d_devs.ptr)
        for b in range(plan.count):
            self._check_devs(devs[b * n_comp:(b + 1) * n_comp])
        if blocks is not None:
            res = self._layout_result(blocks, 0, shots)
        else:
            res = out[:shots]
            res = res if (direct_packed or packed_columns is not None) else res.view(np.bool_)
        if post is not None:
            gone = np.empty(total, dtype=np.uint8)
            hp.d2h(gone, d_gone.ptr)
            return res, ref, gone[:shots].view(np.bool_)
        return res, ref

    def _device_plain(self, shots: int, batch_size: int | None, want_ref: bool, packed_columns: int | None = None,
                      layout: list | None = None):
        """noise -> f -> ``sample_program`` -> layout conversion, the GPU busy beside the channel sampler.

        ``noise="device"``: :meth:`_device_noise_plain`.  ``noise="host"`` (the reference's numpy/PCG64 stream, bit for
        bit): the stream is sequential by contract - ``ChannelSampler`` is one dependency chain per batch, 5 ms per 10^6
        shots - so it runs on a WORKER THREAD (the native sampler releases the GIL) into a ring of pinned staging
        buffers, batch after batch in stream order, while this thread uploads batch i (asynchronous copy from the
        pinned buffer), launches its kernels on the pipeline lanes and sends the rows of batch i - 1 to the host
        (asynchronous copy into recycled pinned result memory).  Wall time = the sampler's, plus the last batch's trip.
        The reference row, when wanted, is row 0 of the first batch (``sampler.py:395-404``)."""
        if self._noise != "host":
            return self._device_noise_plain(shots, batch_size, want_ref, packed_columns, layout=layout)
        import queue
        import threading

        hp = self._hip()
        rides = want_ref
        plan = plan_batches(shots, batch_size, self._estimate_batch_size(), reserve_row=rides)
        ref = None
        cs = self._channel_sampler
        num_f, n_out = cs.num_f, int(self._program.num_outputs)
        wf, wo = max(1, (num_f + 63) // 64), (n_out + 63) // 64
        n_comp = max(1, len(self._program.components))
        total, size = plan.size * plan.count, plan.size
        lanes = self._lane_buffers(hp, size * wf * 8)
        # all columns wanted bit-packed: the kernels write that layout themselves (no padded rows, no compaction pass)
        direct_packed = layout is None and packed_columns is not None and packed_columns == n_out
        row_bytes = (n_out + 7) // 8 if direct_packed else wo * 8
        d_rows, d_devs = self._scratch(hp, "rows", total * row_bytes + 16), self._scratch(hp, "devs", plan.count * n_comp * 4)
        s_up, s_copy = hp.aux_stream(0), hp.aux_stream(1)
        pool = result_pool()
        blocks = None
        if layout is not None:
            blocks = self._layout_blocks(hp, layout, total)
            out = None
        elif direct_packed:
            out = pool.take((total, row_bytes))
        elif packed_columns is not None:
            rb = (packed_columns + 7) // 8
            out = pool.take((total, rb))
            d_c = self._scratch(hp, "compact", total * rb + 16)
        else:
            out = pool.take((total, n_out))
            d_u8 = self._scratch(hp, "unpacked", total * n_out)
        # staging ring: three pinned buffers (one being filled, one being uploaded, one spare)
        n_stage = min(3, plan.count)
        if self._stage is None or self._stage[0].shape != (size, wf) or len(self._stage) < n_stage:
            self._stage = [alloc_pinned_numpy(size * wf * 8, np.uint64, (size, wf)) for _ in range(3)]
        free_q: queue.Queue = queue.Queue()
        full_q: queue.Queue = queue.Queue()
        for i in range(n_stage):
            free_q.put(i)

        def produce() -> None:
            try:
                for b in range(plan.count):
                    i = free_q.get()
                    if i is None:
                        return
                    cs.sample_packed(size, out=self._stage[i])
                    full_q.put(i)
            except BaseException as exc:  # hand the error to the consumer
                full_q.put(exc)

        worker = threading.Thread(target=produce, name="tsim-channel-sampler", daemon=True)
        worker.start()

        def download(b: int) -> None:
            hp.sample_batch_device_end(b % _LANES, s_copy)  # the copy stream waits for exactly this batch
            r0, r1 = b * size, (b + 1) * size
            if blocks is not None:
                self._layout_download(hp, blocks, d_rows.ptr + r0 * row_bytes, wo, r0, r1, s_copy)
            elif direct_packed:
                hp.d2h_async(out[r0:r1], d_rows.ptr + r0 * row_bytes, s_copy)
            elif packed_columns is not None:
                hp.compact_rows_device(d_rows.ptr + r0 * row_bytes, size, packed_columns, d_c.ptr + r0 * rb, in_words=wo, stream=s_copy)
                hp.d2h_async(out[r0:r1], d_c.ptr + r0 * rb, s_copy)
            else:
                hp.unpack_bits_device(d_rows.ptr + r0 * row_bytes, size, n_out, d_u8.ptr + r0 * n_out, stream=s_copy)
                hp.d2h_async(out[r0:r1], d_u8.ptr + r0 * n_out, s_copy)

        try:
            for b in range(plan.count):
                i = full_q.get()
                if isinstance(i, BaseException):
                    raise i
                rows = self._stage[i]
                if rides and b == 0:
                    rows[0] = 0
                lane = b % _LANES
                # the lane's previous launch no longer reads its f buffer (a non-consuming wait: download() joined that
                # launch on the copy stream long ago, sample_batch_device_end would add nothing here)
                hp.pipeline_wait_slot(lane, s_up)
                hp.h2d_async(lanes[lane].ptr, rows, s_up)
                hp.pipeline_wait_stream(s_up)  # every lane is behind the upload
                hp.sample_batch_device_begin(lane, lanes[lane].ptr, size, num_f, self._next_key(),
                                             d_rows.ptr + b * size * row_bytes, d_norm_dev=d_devs.ptr + b * n_comp * 4,
                                             inputs_ready=True, out_bit_packed=direct_packed)
                if b > 0:
                    download(b - 1)
                hp.stream_synchronize(s_up)  # (0.15 ms per 8 MB) the staging buffer may be refilled
                free_q.put(i)
            download(plan.count - 1)
        except BaseException:
            try:  # asynchronous copies into pooled pinned arrays may still be in flight: drain them before the arrays go back
                hp.stream_synchronize(s_copy)
            except Exception:
                pass
            raise
        finally:
            free_q.put(None)
            worker.join()
        for lane in range(min(_LANES, plan.count)):
            hp.sample_batch_device_end(lane)
        hp.stream_synchronize(s_copy)  # behind every batch: the devs below are final
        devs = np.zeros(plan.count * n_comp, dtype=np.float32)
        hp.d2h(devs, d_devs.ptr)
        skip = 1 if rides else 0
        if rides:
            if blocks is not None:  # (arranged output: the caller computed the reference row beforehand, see sample())
                ref = None
            elif packed_columns is None:
                ref = out[0].view(np.bool_).copy()
            else:  # the reference row is wanted as booleans whatever the layout of the rows: one more tiny download
                one = np.zeros((1, n_out), dtype=np.uint8)
                d_one = self._scratch(hp, "ref_row", n_out)
                if direct_packed:  # the kernels wrote bit_packed rows: unpack row 0 on the host
                    ref = np.unpackbits(out[0], bitorder="little")[:n_out].astype(np.bool_)
                else:
                    hp.unpack_bits_device(d_rows.ptr, 1, n_out, d_one.ptr)
                    hp.d2h(one, d_one.ptr)
                    ref = one[0].view(np.bool_).copy()
        for b in range(plan.count):
            self._check_devs(devs[b * n_comp:(b + 1) * n_comp])
        if blocks is not None:
            return self._layout_result(blocks, skip, skip + shots), ref
        res = out[skip:skip + shots]
        return (res if packed_columns is not None else res.view(np.bool_)), ref

    # -- post-selection -----------------------------------------------------------------------------------------
    def _sample_batches_with_postselection(self, shots: int, batch_size: int | None, *, postselection_mask: np.ndarray,
                                           compute_reference: bool = False, xor_detector_ref: bool = False,
                                           packed_columns: int | None = None, xor_observable_ref: bool = False):
        """``(rows, reference or None, was_discarded)``: shots in which a masked, directly readable detector
        fires never reach ``sample_program``; they keep their direct detector columns and False elsewhere
        (sampler.py:422-545)."""
        _check_request(shots, batch_size)
        n_out, nd = int(self._program.num_outputs), self._num_detectors
        if shots == 0:
            return (np.empty((0, n_out), dtype=np.bool_), np.zeros(n_out, dtype=np.bool_) if compute_reference else None,
                    np.empty(0, dtype=np.bool_))
        if not self._program.components:
            rows = self._sample_direct(shots)
            ref = self._compute_reference_sample() if compute_reference else None
            if ref is not None and xor_detector_ref:
                rows[:, :nd] ^= ref[:nd]
            return rows, ref, np.zeros(shots, dtype=np.bool_)
        test_mask = np.asarray(postselection_mask, dtype=np.bool_) & self._direct_detector_mask
        if self._noise == "device" and not self._seam_replaced():
            return self._device_noise_postselect(shots, batch_size, test_mask, compute_reference, xor_detector_ref,
                                                 packed_columns=packed_columns, xor_observable_ref=xor_observable_ref)
        size = plan_batches(shots, batch_size, self._estimate_batch_size()).size
        ref = self._compute_reference_sample() if compute_reference else None
        ref_det = ref[:nd] if (ref is not None and xor_detector_ref) else None
        work = (_SeamPostselect(self, shots, test_mask, ref_det) if self._seam_replaced()
                else _DevicePostselect(self, shots, size, test_mask, ref_det))
        on_device = isinstance(work, _DevicePostselect)
        try:
            if on_device:
                _run_postselected_device(shots, size, work, self._next_key, self._channel_sampler.sample_packed)
            else:
                _run_postselected(shots, size, work, self._next_key)
            if on_device:
                direct = self._direct_detector_mask
                post = self._postselect_masks(test_mask, ref, ref_det, xor_observable_ref)
                rows, gone = work.collect_device(post, packed_columns)
            else:
                rows, gone = work.collect()
        finally:
            if hasattr(work, "release"):
                work.release()
        if not on_device:
            self._apply_detector_reference(rows, gone, ref_det)
            if xor_observable_ref and ref is not None:
                rows[~gone, nd:] ^= ref[nd:]
            if packed_columns is not None:
                rows = np.packbits(rows[:, :packed_columns], axis=1, bitorder="little")
        return rows, ref, gone

    def _postselect_masks(self, test_mask, ref, ref_det, xor_observable_ref: bool) -> dict:
        """The five column masks of ``tsim_postselect_rows_device`` (sampler.py:532-540 and the reference-sample flags)."""
        nd, n_out = self._num_detectors, int(self._program.num_outputs)

        def cols(det_bits=None, obs_bits=None) -> np.ndarray:
            full = np.zeros(n_out, dtype=np.uint8)
            if det_bits is not None:
                full[:nd] = np.asarray(det_bits, dtype=np.uint8)
            if obs_bits is not None:
                full[nd:] = np.asarray(obs_bits, dtype=np.uint8)
            return full

        direct = self._direct_detector_mask
        return {
            "test": cols(test_mask),
            "test_ref": cols(ref_det),
            "keep": cols(direct),
            "xor_kept": cols(ref_det, ref[nd:] if (ref is not None and xor_observable_ref) else None),
            "xor_discarded": cols(ref_det & direct if ref_det is not None else None),
        }

    def _apply_detector_reference(self, rows: np.ndarray, gone: np.ndarray, ref_det: np.ndarray | None) -> None:
        """XOR the reference's detector bits in: all of them for shots that ran, only the directly readable
        ones for discarded shots (their other columns are blank, sampler.py:532-540)."""
        if ref_det is None:
            return
        nd = self._num_detectors
        rows[~gone, :nd] ^= ref_det
        rows[gone, :nd] ^= ref_det & self._direct_detector_mask

    def _device_noise_postselect(self, shots, batch_size, test_mask, compute_reference, xor_detector_ref, *,
                                 packed_columns: int | None = None, xor_observable_ref: bool = False):
        """``noise="device"``: there is no host stream that says which shots reach ``sample_program`` - so EVERY row goes
        through the fast sampling path (:meth:`_device_noise_plain`: noise, fused first passes, downloads as a pipeline) and
        one small kernel blanks the rows in which a masked direct detector fires, exactly as the reference returns them
        (their direct detector columns, False elsewhere; reference bits as in ``sampler.py:532-540``).  Sampling a
        discarded row costs 11 ns per 10^3 shots; compacting the survivors first (gather, dense batches, scatter, a host
        round trip per chunk) cost 125x the plain path.  ``packed_columns``: return ``bit_packed`` rows of that many
        leading columns instead of bools; ``xor_observable_ref``: also XOR the reference's observable bits into the
        surviving rows (``sample(use_observable_reference_sample=True)``)."""
        nd = self._num_detectors
        ref = self._compute_reference_sample() if compute_reference else None
        ref_det = ref[:nd] if (ref is not None and xor_detector_ref) else None
        post = self._postselect_masks(test_mask, ref, ref_det, xor_observable_ref)
        rows, _, gone = self._device_noise_plain(shots, batch_size, False, packed_columns, post=post)
        return rows, ref, gone

    def __repr__(self) -> str:
        """Compilation statistics in the reference's format (sampler.py:557-609)."""
        levels = [(len(c.output_indices), lv) for c in self._program.components for lv in c.compiled_scalar_graphs]

        def size_of(*arrays) -> int:
            return int(sum(np.asarray(a).size for a in arrays))

        terms = {
            "A": sum(size_of(lv.node_phases.phases) for _, lv in levels),
            "B": sum(size_of(lv.halfpi_phases.coeffs) for _, lv in levels),
            "C": sum(size_of(lv.pi_products.psi_const) for _, lv in levels),
            "D": sum(size_of(lv.phase_pairs.alpha, lv.phase_pairs.beta) for _, lv in levels),
        }
        nbytes = sum(np.asarray(v).nbytes for _, lv in levels
                     for fam in (lv.node_phases, lv.halfpi_phases, lv.pi_products, lv.phase_pairs, lv.prefactor)
                     for v in vars(fam).values() if isinstance(v, np.ndarray))
        unit, scale = ("B", 1) if nbytes < 1024 else ("kB", 1024) if nbytes < 1024**2 else ("MB", 1024**2)
        mem = f"{nbytes} B" if scale == 1 else f"{nbytes / scale:.1f} {unit}"
        return (
            f"{type(self).__name__}({len(self._program.direct_f_indices)} direct, "
            f"{sum(lv.num_graphs for _, lv in levels)} graphs, "
            f"{sum(ch.num_bits for ch in self._channel_sampler.channels)} error channel bits, "
            f"{max((n for n, _ in levels), default=0)} outputs for largest cc, "
            f"≤ {max((lv.n_params for _, lv in levels), default=0)} parameters, "
            f"{terms['A']} A terms, {terms['B']} B terms, {terms['C']} C terms, {terms['D']} D terms, {mem})"
        )


class CompiledMeasurementSampler(_CompiledSamplerBase):
    """Samples measurement outcomes (sequential levels 0..n per component)."""

    def sample(self, shots: int, *, batch_size: int | None = None) -> np.ndarray:
        return self._sample_batches(shots, batch_size)


def _deliver(bit_packed: bool, *blocks: np.ndarray):
    """The column blocks as the caller gets them: bools, or - ``bit_packed=True`` - little-endian bytes per row
    (sampler.py:665-669); one block comes back bare, several as a tuple."""
    if bit_packed:
        blocks = tuple(np.packbits(b.view(np.uint8) if b.dtype == np.bool_ else b, axis=1, bitorder="little") for b in blocks)
    return blocks[0] if len(blocks) == 1 else blocks


class CompiledDetectorSampler(_CompiledSamplerBase):
    """Samples detector and observable outcomes."""

    def sample(self, shots: int, *, batch_size: int | None = None, prepend_observables: bool = False,
               append_observables: bool = False, separate_observables: bool = False, bit_packed: bool = False,
               use_detector_reference_sample: bool = False, use_observable_reference_sample: bool = False,
               postselection_mask: np.ndarray | None = None):
        """Detector samples with the reference's column-arrangement, reference-sample and post-selection flags
        (sampler.py:732-868)."""
        if separate_observables and (prepend_observables or append_observables):
            raise ValueError("Can't specify separate_observables=True with append_observables=True or prepend_observables=True")
        nd, n_out = self._num_detectors, int(self._program.num_outputs)
        want_ref = use_detector_reference_sample or use_observable_reference_sample
        mask = None
        if postselection_mask is not None:
            mask = np.asarray(postselection_mask, dtype=np.bool_)
            if mask.shape != (nd,):
                raise ValueError(f"postselection_mask must have shape ({nd},), got {mask.shape}")
            if not self._program.components or not (mask & self._direct_detector_mask).any():
                mask = None  # no shot could be skipped: the plain path does the same work

        if mask is not None:
            width = n_out if append_observables else nd
            fast = (bit_packed and not separate_observables and not prepend_observables and width > 0 and shots > 0
                    and not self._seam_replaced())
            # bit_packed rows of a prefix of the columns: blanking, reference bits and packing all happen on the device
            rows, ref, gone = self._sample_batches_with_postselection(
                shots, batch_size, postselection_mask=mask, compute_reference=want_ref,
                xor_detector_ref=use_detector_reference_sample, packed_columns=width if fast else None,
                xor_observable_ref=use_observable_reference_sample)
            if fast:
                return rows
        elif (shots > 0 and self._program.components and not self._seam_replaced()
              and (want_ref or prepend_observables or (separate_observables and bit_packed))):
            # everything sample() does to the rows - column blocks, their order, the reference-sample flips, bit-packing -
            # happens on the device (tsim_arrange_rows_device); the host receives the arrays it hands out
            _check_request(shots, batch_size)
            ref = None
            if want_ref:
                if self._noise == "device":
                    ref = self._compute_reference_sample()  # (its own key, then the batches: as _device_noise_plain does it)
                else:
                    # host noise: the reference row rides as row 0 of the first batch (sampler.py:395-404) - the same row, same
                    # subkey, same in-batch index computed here first, so that its bits are known before the rows are arranged
                    quiet = np.zeros((1, self._channel_sampler.num_f), dtype=np.uint8)
                    ref = np.asarray(self._seam(quiet, prng.split(self._key)[1])[0], dtype=np.bool_)
            det = np.arange(nd, dtype=np.uint32)
            obs = np.arange(nd, n_out, dtype=np.uint32)
            if ref is not None and use_detector_reference_sample:
                det = det | (ref[:nd].astype(np.uint32) << np.uint32(31))
            if ref is not None and use_observable_reference_sample:
                obs = obs | (ref[nd:].astype(np.uint32) << np.uint32(31))
            if separate_observables:
                cols = [det, obs]
            elif prepend_observables:
                cols = [np.concatenate([obs, det] + ([obs] if append_observables else []))]
            else:
                cols = [np.concatenate([det, obs]) if append_observables else det]
            rides = want_ref and self._noise != "device"
            res = self._device_plain(shots, batch_size, rides, layout=[(c, bit_packed) for c in cols])[0]
            return tuple(res) if separate_observables else res[0]
        elif want_ref:
            rows, ref = self._sample_batches(shots, batch_size, compute_reference=True)
            flip = np.zeros(n_out, dtype=np.bool_)
            if use_detector_reference_sample:
                flip[:nd] = ref[:nd]
            if use_observable_reference_sample:
                flip[nd:] = ref[nd:]
            rows = rows ^ flip
        else:
            width = n_out if append_observables else nd
            if (bit_packed and not separate_observables and not prepend_observables
                    and width > 0 and shots > 0 and self._program.components and not self._seam_replaced()):
                # the wanted columns are a prefix of the packed device rows: compact there, move width/8 bytes per shot
                _check_request(shots, batch_size)
                return self._device_plain(shots, batch_size, False, packed_columns=width)[0]
            if (bit_packed and not separate_observables and not prepend_observables and width > 0 and shots > 0
                    and not self._program.components and self._direct_on_device(shots)):
                _check_request(shots, batch_size)
                return self._direct_device(shots, batch_size, packed_columns=width)
            rows = self._sample_batches(shots, batch_size)

        det, obs = rows[:, :nd], rows[:, nd:]
        if separate_observables:
            return _deliver(bit_packed, det, obs)
        if not prepend_observables:  # detectors first: a prefix of the rows as they are (no 80 MB copy)
            return _deliver(bit_packed, rows if append_observables else det)
        return _deliver(bit_packed, np.concatenate([obs, det] + ([obs] if append_observables else []), axis=1))


class CompiledStateProbs(_CompiledSamplerBase):
    """``P(state | error sample)`` from joint-mode programs (levels [0, n] per component)."""

    def probability_of(self, state: np.ndarray, *, batch_size: int) -> np.ndarray:
        """``p_joint / p_norm`` per sampled error configuration (sampler.py:906-953)."""
        if batch_size < 1:
            raise ValueError(f"batch_size must be at least 1, got {batch_size}")
        state = np.asarray(state)
        n_out = int(self._program.num_outputs)
        if state.shape != (n_out,):
            raise ValueError(f"state must have shape ({n_out},), got {state.shape}")
        f = self._channel_sampler.sample(batch_size)
        d = self._direct
        agree = np.ones(batch_size, dtype=np.float32)
        if len(d.f_index):
            want = state[d.column].astype(np.bool_)
            agree = ((f[:, d.f_index].astype(np.bool_) ^ d.flip) == want).all(axis=1).astype(np.float32)
        p_norm, p_joint = np.ones(batch_size, dtype=np.float32), agree
        hp = self._hip()
        for ci, comp in enumerate(self._program.components):
            if len(comp.compiled_scalar_graphs) != 2:
                raise ValueError("probability_of needs a joint-mode program (two levels per component)")
            f_sel = f[:, np.asarray(comp.f_selection, dtype=np.int64)]
            plugged = np.broadcast_to(state[list(comp.output_indices)].astype(np.uint8), (batch_size, len(comp.output_indices)))
            p_norm = p_norm * hp.evaluate(ci, 0, f_sel, return_abs=True)
            p_joint = p_joint * hp.evaluate(ci, 1, np.hstack([f_sel, plugged]), return_abs=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.asarray(p_joint / p_norm)
