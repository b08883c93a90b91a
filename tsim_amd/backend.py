"""HIP backend: the reference's L3 functions, served by ``libtsim_hip.so``.

Mirrors, with the same names, argument meaning and error behaviour:

* ``sample_program(program, f_params, key) -> bool[B, num_outputs]``
  (reference: src/tsim/sampler.py:117-167) - raises ``ValueError`` on a
  vanishing marginal (normalisation deviation ~ 1) and ``warnings.warn`` s above
  1e-5 (sampler.py:149-161);
* ``evaluate(circuit, param_vals) -> complex64[B]``
  (reference: src/tsim/compile/evaluate.py:15-59).

Host orchestration only: ctypes + numpy, no torch, no jax.  All arithmetic is in
the HIP kernels; if the extension is missing the import of ``_lib`` raises.
"""

from __future__ import annotations

import ctypes as C
import warnings
import weakref

import math

import numpy as np

from . import _lib
from .program import CompiledProgram, CompiledScalarGraphs, from_tsim, validate_program


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _level_desc(lv: CompiledScalarGraphs, keep: list) -> _lib.LevelDesc:
    """Fill a ``tsim_level_desc`` with pointers into C-contiguous copies (kept alive in ``keep``)."""
    a, b, c, d, pre = lv.node_phases, lv.halfpi_phases, lv.pi_products, lv.phase_pairs, lv.prefactor
    G, P = int(lv.num_graphs), int(lv.n_params)
    arrs = dict(
        a_phases=_c(a.phases, np.uint8),
        a_params=_c(a.params, np.uint8),
        a_counts=_c(a.counts, np.int32),
        b_coeffs=_c(b.coeffs, np.uint8),
        b_params=_c(b.params, np.uint8),
        c_psi_const=_c(c.psi_const, np.uint8),
        c_psi_params=_c(c.psi_params, np.uint8),
        c_phi_const=_c(c.phi_const, np.uint8),
        c_phi_params=_c(c.phi_params, np.uint8),
        d_alpha=_c(d.alpha, np.uint8),
        d_alpha_params=_c(d.alpha_params, np.uint8),
        d_beta=_c(d.beta, np.uint8),
        d_beta_params=_c(d.beta_params, np.uint8),
        d_counts=_c(d.counts, np.int32),
        phase_indices=_c(pre.phase_indices, np.uint8),
        floatfactor=_c(pre.floatfactor, np.int32).reshape(-1, 4),
        power2=_c(pre.power2, np.int32),
        approx=_c(pre.approximate_floatfactors, np.complex64),
    )

    def dim1(x):
        return int(x.shape[1]) if x.ndim >= 2 else 0

    desc = _lib.LevelDesc()
    desc.num_graphs, desc.n_params = G, P
    desc.ta = dim1(arrs["a_phases"]) if G else 0
    desc.tb = dim1(arrs["b_coeffs"]) if G else 0
    desc.tc = dim1(arrs["c_psi_const"]) if G else 0
    desc.td = dim1(arrs["d_alpha"]) if G else 0
    # shape checks the C side cannot do (it only sees pointers)
    want = {
        "a_phases": (G, desc.ta), "a_params": (G, desc.ta, P), "a_counts": (G,),
        "b_coeffs": (G, desc.tb), "b_params": (G, desc.tb, P),
        "c_psi_const": (G, desc.tc), "c_psi_params": (G, desc.tc, P),
        "c_phi_const": (G, desc.tc), "c_phi_params": (G, desc.tc, P),
        "d_alpha": (G, desc.td), "d_alpha_params": (G, desc.td, P),
        "d_beta": (G, desc.td), "d_beta_params": (G, desc.td, P), "d_counts": (G,),
        "phase_indices": (G,), "floatfactor": (G, 4), "power2": (G,), "approx": (G,),
    }
    for name, shape in want.items():
        if G and arrs[name].size != math.prod(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {arrs[name].shape}")
    for name, arr in arrs.items():
        keep.append(arr)
        # (the address without a ctypes view object per array: a fresh C2 handle is 2 ms, 18 arrays x 6 levels of them were 0.1)
        setattr(desc, name, arr.__array_interface__["data"][0] if arr.size else None)
    desc.has_approx = 1 if pre.has_approximate_floatfactors else 0
    return desc


class _Handle:
    """The ``tsim_program*`` and what hangs off it, shared by a :class:`HipProgram` and the objects
    that borrow it (:class:`DeviceBuffer`, :class:`DeviceNoiseSampler`).  ``close`` runs once - from
    ``HipProgram.close()``, its finalizer or interpreter exit - destroys the dependent noise handles,
    then the program (which frees every buffer still allocated on it) and clears ``h``: borrowers
    check ``h`` and never touch a destroyed handle, whatever order garbage collection picks."""

    def __init__(self, lib, h):
        self.lib = lib
        self.h = h
        self.noise: list = []  # live tsim_noise* handles

    def close(self) -> None:
        h, self.h = self.h, None
        if not h:
            return
        for n in self.noise:
            self.lib.tsim_noise_destroy(n)
        self.noise.clear()
        self.lib.tsim_program_destroy(h)


class DeviceBuffer:
    """A ``hipMalloc`` allocation owned by a :class:`HipProgram` (freed with it at the latest)."""

    def __init__(self, prog: "HipProgram", nbytes: int):
        self._handle = prog._handle
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _lib.check(prog._lib.tsim_malloc_device(prog._h, self.nbytes, C.byref(p)), "tsim_malloc_device")
        self.ptr = p.value or 0

    def free(self) -> None:
        hd = self._handle
        if self.ptr and hd.h:
            hd.lib.tsim_free_device(hd.h, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.free()
        except Exception:
            pass


class HipProgram:
    """A compiled program uploaded to one MI355X (handle over ``tsim_program*``)."""

    def __init__(self, program, device: int = 0, mode: str = "auto", pattern_tables=None):
        """``mode``: "auto" (exact-value fast formulation when every graph qualifies) or
        "faithful" (operation-by-operation int32 mirror of the reference) or "rows" (exact-value
        formulation on the row-by-row kernel instead of the LDS chunk-table kernel); see
        include/tsim_hip.h.

        ``pattern_tables``: None = default (low-weight error-pattern tables on in "auto" mode),
        False = off, True = on, or an int 0..7 = on with exactly that maximum tabulated weight (no on-demand deepening)."""
        if mode not in ("auto", "faithful", "rows"):
            raise ValueError("mode must be 'auto', 'faithful' or 'rows'")
        self._lib = _lib.load()
        self._handle = _Handle(self._lib, None)
        program = from_tsim(program)
        validate_program(program)
        self.program = program
        self.device = int(device)
        self.num_outputs = int(program.num_outputs)
        self.n_components = len(program.components)
        lib = self._lib
        dfi = _c(program.direct_f_indices, np.int32)
        flips = _c(np.asarray(program.direct_flips).astype(np.uint8), np.uint8)
        order = _c(program.output_order, np.int32)
        h = C.c_void_p()
        _lib.check(
            lib.tsim_program_create(
                self.num_outputs, int(program.num_detectors), len(dfi),
                _lib.ptr(dfi), _lib.ptr(flips), _lib.ptr(order), C.byref(h),
            ),
            "tsim_program_create",
        )
        self._handle.h = h
        self._finalizer = weakref.finalize(self, _Handle.close, self._handle)
        try:
            for comp in program.components:
                oi = _c(comp.output_indices, np.int32)
                fs = _c(comp.f_selection, np.int32)
                ci = _lib.check(
                    lib.tsim_program_add_component(
                        h, len(oi), _lib.ptr(oi), len(fs), _lib.ptr(fs), len(comp.compiled_scalar_graphs)
                    ),
                    "tsim_program_add_component",
                )
                for lv in comp.compiled_scalar_graphs:
                    keep: list = []
                    desc = _level_desc(lv, keep)
                    _lib.check(lib.tsim_program_add_level(h, ci, C.byref(desc)), "tsim_program_add_level")
            _lib.check(lib.tsim_program_set_mode(h, {"auto": 0, "faithful": 1, "rows": 2}[mode]), "tsim_program_set_mode")
            if pattern_tables is not None:
                if isinstance(pattern_tables, bool):
                    en, cap = (1 if pattern_tables else 0), -1
                else:
                    en, cap = 1, int(pattern_tables)
                _lib.check(lib.tsim_program_set_pattern_tables(h, en, cap), "tsim_program_set_pattern_tables")
            _lib.check(lib.tsim_program_finalize(h, self.device), "tsim_program_finalize")
            fast = C.c_int32(0)
            _lib.check(lib.tsim_program_get_mode(h, C.byref(fast)), "tsim_program_get_mode")
            self.fast = bool(fast.value)
        except Exception:
            self.close()
            raise
        self._split_buf = (C.c_uint32 * 4)()
        self._packed_bufs = None

    @property
    def _h(self):
        """The live ``tsim_program*``; raises once the handle has been closed."""
        h = self._handle.h
        if not h:
            raise _lib.HipBackendError("HipProgram is closed")
        return h

    def close(self) -> None:
        """Destroy the device image, every buffer allocated on it and its noise samplers (idempotent)."""
        self._finalizer()

    # -- info ---------------------------------------------------------------
    def info(self) -> dict:
        nc, no = C.c_int32(), C.c_int32()
        ib, tg, tr = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(
            self._lib.tsim_program_info(self._h, C.byref(nc), C.byref(no), C.byref(ib), C.byref(tg), C.byref(tr)),
            "tsim_program_info",
        )
        st = (C.c_int64 * 8)()
        _lib.check(self._lib.tsim_program_stats(self._h, st), "tsim_program_stats")
        en, tb = C.c_int32(), C.c_int64()
        mw = (C.c_int32 * max(1, nc.value))()
        _lib.check(self._lib.tsim_program_pattern_table_info(self._h, C.byref(en), C.byref(tb), mw),
                   "tsim_program_pattern_table_info")
        pend = C.c_int32()
        _lib.check(self._lib.tsim_program_tables_pending(self._h, C.byref(pend)), "tsim_program_tables_pending")
        return dict(n_components=nc.value, num_outputs=no.value, image_bytes=ib.value, pattern_build_pending=bool(pend.value),
                    total_graphs=tg.value, total_rows=tr.value, fast=bool(st[0]), levels=st[1],
                    fixed_frame_levels=st[2], product_pairs=st[3], counted_rows=st[4],
                    table_bytes=st[5], graphs_d_tabled=st[6], chunk_table_kernel=((st[7] & 15) == 1), wide_sparse_kernel=((st[7] & 15) == 2),
                    wide_fused_kernel=bool(st[7] & 32), wide_shared_columns=bool(st[7] & 16), reference_sum_wrap_possible=bool(st[7] & 64),
                    pattern_tables=bool(en.value), pattern_table_bytes=tb.value,
                    pattern_max_weight=[int(mw[i]) for i in range(nc.value)])

    PATH_NAMES = ("lw_fast", "lw_fastm", "lw_multi", "direct_multi", "wide", "lw_fast1", "lw_reg", "lw_lds", "lw_lds_wide",
                  "sample4w", "sample4", "sample4h", "hw", "over", "rows", "sample4h_multi", "gen", "noise_fast")

    def path_counts(self, reset: bool = False) -> dict:
        """Launches per kernel family since creation / the last reset (``tsim_program_path_counts``), non-zero entries only."""
        n = 24
        out = (C.c_int64 * n)()
        _lib.check(self._lib.tsim_program_path_counts(self._h, out, 1 if reset else 0), "tsim_program_path_counts")
        return {name: int(out[i]) for i, name in enumerate(self.PATH_NAMES) if out[i]}

    # -- the hot path, host buffers -------------------------------------------
    def sample_batch(self, f_params: np.ndarray, key, *, shot_offset: int = 0, bit_packed: bool = False):
        """One batch through the fused kernel.  Returns ``(samples, max_norm_dev[n_components])``."""
        f = np.asarray(f_params)
        if f.ndim != 2:
            raise ValueError(f"f_params must be 2-D (batch, num_f), got shape {f.shape}")
        if f.dtype != np.uint8:
            f = (f != 0).astype(np.uint8)
        f = np.ascontiguousarray(f)
        B, num_f = f.shape
        wo = (self.num_outputs + 63) // 64
        if bit_packed:
            out = np.zeros((B, wo * 8), dtype=np.uint8)
        else:
            out = np.zeros((B, self.num_outputs), dtype=np.uint8)
        devs = np.zeros(max(1, self.n_components), dtype=np.float32)
        _lib.check(
            self._lib.tsim_sample_batch(
                self._h, _lib.ptr(f), B, num_f, int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF,
                int(shot_offset), _lib.ptr(out), 1 if bit_packed else 0, _lib.ptr(devs),
            ),
            "tsim_sample_batch",
        )
        if not bit_packed:
            out = out.view(np.bool_)
        return out, devs[: self.n_components]

    def sample_batch_packed(self, f_packed: np.ndarray, num_f: int, key, *, shot_offset: int = 0):
        """One batch from PACKED error rows (``uint64[B, ceil(num_f/64)]``, e.g.
        ``ChannelSampler.sample_packed``): 8 bytes per 64 error bits over PCIe instead of one byte per
        bit.  Returns ``(bool[B, num_outputs], max_norm_dev[n_components])`` like :meth:`sample_batch`."""
        f = np.ascontiguousarray(f_packed, dtype=np.uint64)
        wf = max(1, (int(num_f) + 63) // 64)
        if f.ndim != 2 or f.shape[1] != wf:
            raise ValueError(f"f_packed must have shape (batch, {wf}), got {f.shape}")
        B = f.shape[0]
        wo = (self.num_outputs + 63) // 64
        out = np.empty((B, self.num_outputs), dtype=np.uint8)
        devs = np.zeros(max(1, self.n_components), dtype=np.float32)
        if B == 0 or self.num_outputs == 0:
            return out.view(np.bool_), devs[: self.n_components]
        need = (B * wf * 8, B * wo * 8, B * self.num_outputs)
        bufs = self._packed_bufs
        if bufs is None or any(b.nbytes < n for b, n in zip(bufs[:3], need)):
            bufs = self._packed_bufs = (self.malloc(need[0]), self.malloc(need[1]), self.malloc(need[2]),
                                        self.malloc(4 * max(1, self.n_components)))
        d_f, d_o, d_u8, d_dev = bufs
        self.h2d(d_f, f)
        self.sample_batch_device(d_f.ptr, B, num_f, key, d_o.ptr, shot_offset=shot_offset, d_norm_dev=d_dev.ptr)
        self.unpack_bits_device(d_o.ptr, B, self.num_outputs, d_u8.ptr)
        self.d2h(out, d_u8.ptr)
        self.d2h(devs, d_dev.ptr)
        return out.view(np.bool_), devs[: self.n_components]

    # -- the hot path, device-resident ----------------------------------------
    def malloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def h2d(self, dst: DeviceBuffer | int, src: np.ndarray) -> None:
        src = np.ascontiguousarray(src)
        d = dst.ptr if isinstance(dst, DeviceBuffer) else int(dst)
        _lib.check(self._lib.tsim_memcpy_h2d(self._h, C.c_void_p(d), _lib.ptr(src), src.nbytes), "tsim_memcpy_h2d")

    def d2h(self, dst: np.ndarray, src: DeviceBuffer | int) -> None:
        assert dst.flags.c_contiguous
        s = src.ptr if isinstance(src, DeviceBuffer) else int(src)
        _lib.check(self._lib.tsim_memcpy_d2h(self._h, _lib.ptr(dst), C.c_void_p(s), dst.nbytes), "tsim_memcpy_d2h")

    def pack_bits_device(self, d_in: int, B: int, nbits: int, d_out: int) -> None:
        _lib.check(
            self._lib.tsim_pack_bits_device(self._h, C.c_void_p(d_in), B, nbits, C.c_void_p(d_out), None),
            "tsim_pack_bits_device",
        )

    def unpack_bits_device(self, d_in: int, B: int, nbits: int, d_out: int, stream: int = 0) -> None:
        _lib.check(
            self._lib.tsim_unpack_bits_device(self._h, C.c_void_p(d_in), B, nbits, C.c_void_p(d_out), stream or None),
            "tsim_unpack_bits_device",
        )

    def mem_info(self) -> tuple[int, int]:
        """``(free, total)`` bytes of the handle's device."""
        fr, tot = C.c_int64(), C.c_int64()
        _lib.check(self._lib.tsim_mem_info(self._h, C.byref(fr), C.byref(tot)), "tsim_mem_info")
        return int(fr.value), int(tot.value)

    def gather_rows_device(self, d_src: int, words: int, d_index: int, n_valid: int, n_total: int, d_dst: int) -> None:
        """``dst[i] = src[index[i if i < n_valid else 0]]`` for ``i < n_total`` (rows of ``words`` uint64)."""
        _lib.check(
            self._lib.tsim_gather_rows_device(self._h, C.c_void_p(d_src), int(words), C.c_void_p(d_index), int(n_valid),
                                              int(n_total), C.c_void_p(d_dst), None),
            "tsim_gather_rows_device",
        )

    def scatter_rows_device(self, d_src: int, words: int, d_index: int, n: int, d_dst: int) -> None:
        """``dst[index[i]] = src[i]`` for ``i < n``."""
        _lib.check(
            self._lib.tsim_scatter_rows_device(self._h, C.c_void_p(d_src), int(words), C.c_void_p(d_index), int(n),
                                               C.c_void_p(d_dst), None),
            "tsim_scatter_rows_device",
        )

    def compact_rows_device(self, d_in: int, B: int, nbits: int, d_out: int, *, in_words: int = 0, stream: int = 0) -> None:
        """Padded uint64 rows (``in_words`` per row, default ``ceil(nbits/64)``) -> ``ceil(nbits/8)``-byte
        rows of the first ``nbits`` columns (``np.packbits(bits[:, :nbits], axis=1, bitorder="little")``)."""
        _lib.check(
            self._lib.tsim_compact_rows_device(self._h, C.c_void_p(d_in), int(B), int(in_words), int(nbits),
                                               C.c_void_p(d_out), stream or None),
            "tsim_compact_rows_device",
        )

    def arrange_rows_device(self, d_in: int, B: int, in_words: int, d_cols: int, n_cols: int, packed: bool, d_out: int, *, stream: int = 0) -> None:
        """Column selection / order / constant flips of padded rows, as bytes or bit-packed (``tsim_arrange_rows_device``)."""
        _lib.check(self._lib.tsim_arrange_rows_device(self._h, C.c_void_p(d_in), int(B), int(in_words), C.c_void_p(d_cols), int(n_cols),
                                                      1 if packed else 0, C.c_void_p(d_out), stream or None), "tsim_arrange_rows_device")

    def survivors_append_device(self, d_gone: int, n: int, base: int, d_scratch: int, d_queue: int, d_tail: int) -> None:
        """Append the shot ids ``base + i`` of the rows with ``gone[i] == 0``, in order, to a device queue (``tsim_survivors_append_device``)."""
        _lib.check(self._lib.tsim_survivors_append_device(self._h, C.c_void_p(d_gone), int(n), int(base), C.c_void_p(d_scratch),
                                                          C.c_void_p(d_queue), C.c_void_p(d_tail), None), "tsim_survivors_append_device")

    def postselect_rows_device(self, d_rows: int, B: int, row_bytes: int, d_masks: int, d_gone: int = 0, *, stream: int = 0) -> None:
        """Blank the rows in which a masked direct detector fires (``tsim_postselect_rows_device``, include/tsim_hip.h)."""
        _lib.check(self._lib.tsim_postselect_rows_device(self._h, C.c_void_p(d_rows), int(B), int(row_bytes), C.c_void_p(d_masks),
                                                         C.c_void_p(d_gone) if d_gone else None, stream or None),
                   "tsim_postselect_rows_device")

    def sample_batch_device(self, d_f: int, B: int, num_f: int, key, d_out: int, *,
                            shot_offset: int = 0, d_norm_dev: int = 0, stream: int = 0) -> None:
        """Asynchronous launch on the handle's stream (``stream``: a HIP stream of the caller instead); buffers are raw
        device pointers."""
        _lib.check(
            self._lib.tsim_sample_batch_device(
                self._h, C.c_void_p(d_f), int(B), int(num_f), int(key[0]) & 0xFFFFFFFF,
                int(key[1]) & 0xFFFFFFFF, int(shot_offset), C.c_void_p(d_out),
                C.c_void_p(d_norm_dev) if d_norm_dev else None, C.c_void_p(stream) if stream else None,
            ),
            "tsim_sample_batch_device",
        )

    PIPELINE_SLOTS = 32

    def split_key(self, key):
        """``prng.split`` computed by the library (same values; ~5 us cheaper per batch)."""
        out = self._split_buf
        self._lib.tsim_key_split(key[0], key[1], out)
        return (out[0], out[1]), (out[2], out[3])

    def sample_batch_device_begin(self, slot: int, d_f: int, B: int, num_f: int, key, d_out: int, *,
                                  shot_offset: int = 0, d_norm_dev: int = 0, inputs_ready: bool = False,
                                  out_bit_packed: bool = False) -> None:
        """Pipelined launch on the slot's own stream (launches of different slots overlap).  ``d_out``
        is complete only after ``sample_batch_device_end(slot)``.  ``inputs_ready=True``: ``d_f`` is
        already complete and nothing queued on the handle's stream still uses ``d_out`` - the launch
        then needs no cross-stream event.  ``out_bit_packed=True``: ``d_out`` receives ``uint8[B, ceil(num_outputs/8)]``
        rows (the reference's ``bit_packed`` layout) instead of the padded 8-byte words."""
        rc = self._lib.tsim_sample_batch_device_begin(
            self._h, slot, d_f, B, num_f, key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF, shot_offset, d_out,
            d_norm_dev or None, None, (1 if inputs_ready else 0) | (2 if out_bit_packed else 0))
        if rc < 0:
            _lib.check(rc, "tsim_sample_batch_device_begin")

    def aux_stream(self, index: int) -> int:
        """One of the handle's auxiliary ``hipStream_t`` (noise sampling, transfers beside the sampling lanes)."""
        st = C.c_void_p()
        _lib.check(self._lib.tsim_aux_stream(self._h, int(index), C.byref(st)), "tsim_aux_stream")
        return int(st.value or 0)

    def pipeline_next_slot(self) -> int:
        """The pipeline slot the next batch of :meth:`sample_steps_device` will take."""
        n = C.c_int32(0)
        _lib.check(self._lib.tsim_pipeline_next_slot(self._h, C.byref(n)), "tsim_pipeline_next_slot")
        return int(n.value)

    def d2h_async(self, dst: np.ndarray, src: int, stream: int) -> None:
        assert dst.flags.c_contiguous
        _lib.check(self._lib.tsim_memcpy_d2h_async(self._h, _lib.ptr(dst), C.c_void_p(int(src)), dst.nbytes, stream or None), "tsim_memcpy_d2h_async")

    def h2d_async(self, dst: int, src: np.ndarray, stream: int) -> None:
        assert src.flags.c_contiguous
        _lib.check(self._lib.tsim_memcpy_h2d_async(self._h, C.c_void_p(int(dst)), _lib.ptr(src), src.nbytes, stream or None), "tsim_memcpy_h2d_async")

    def stream_synchronize(self, stream: int) -> None:
        _lib.check(self._lib.tsim_stream_synchronize(self._h, stream or None), "tsim_stream_synchronize")

    def sample_steps_device(self, d_f, B: int, num_f: int, key_state, d_out, *, shot_offset: int = 0,
                            inputs_ready: bool = False, out_bit_packed: bool = False, d_norm_dev=None) -> None:
        """``len(d_f)`` consecutive batches in one call (the reference's batch loop, ``sampler.py:340-420``): for each,
        ``key, subkey = split(key)`` and one ``sample_program``; ``key_state`` is a ``(c_uint32 * 2)`` advanced in
        place.  ``d_f`` / ``d_out``: sequences of device addresses (or ``(c_void_p * n)`` arrays, reused as they are).
        Results are complete after ``synchronize()``; see ``tsim_sample_steps_device`` in include/tsim_hip.h."""
        n = len(d_f)
        fa = d_f if isinstance(d_f, C.Array) else (C.c_void_p * n)(*[int(x) for x in d_f])
        oa = d_out if isinstance(d_out, C.Array) else (C.c_void_p * n)(*[int(x) for x in d_out])
        da = None if d_norm_dev is None else (C.c_void_p * n)(*[int(x) for x in d_norm_dev])
        rc = self._lib.tsim_sample_steps_device(self._h, n, fa, int(B), int(num_f), key_state, int(shot_offset), oa, da,
                                                (1 if inputs_ready else 0) | (2 if out_bit_packed else 0))
        if rc < 0:
            _lib.check(rc, "tsim_sample_steps_device")

    def sample_steps_noise_device(self, noise: "DeviceNoiseSampler", d_f, B: int, num_f: int, key_state, noise_key_state, d_out, *,
                                  shot_offset: int = 0, out_bit_packed: bool = False, d_norm_dev=None) -> None:
        """:meth:`sample_steps_device` with the f rows drawn on the device in the same call (``tsim_sample_steps_noise_device``):
        batch j's noise key is the j-th split of ``noise_key_state`` (a ``(c_uint32 * 2)``, advanced in place); its rows are
        written to ``d_f[j]`` - the bytes ``noise.sample_into`` writes for that key.  One-component programs of at most 8
        outputs over f rows of at most 128 bits draw the noise inside their first pass (one kernel)."""
        n = len(d_f)
        fa = d_f if isinstance(d_f, C.Array) else (C.c_void_p * n)(*[int(x) for x in d_f])
        oa = d_out if isinstance(d_out, C.Array) else (C.c_void_p * n)(*[int(x) for x in d_out])
        da = None if d_norm_dev is None else (C.c_void_p * n)(*[int(x) for x in d_norm_dev])
        rc = self._lib.tsim_sample_steps_noise_device(self._h, noise._n, n, fa, int(B), int(num_f), key_state, noise_key_state, int(shot_offset),
                                                      oa, da, 2 if out_bit_packed else 0)
        if rc < 0:
            _lib.check(rc, "tsim_sample_steps_noise_device")

    def profile_read_steps(self, reset: bool = True) -> int:
        n = C.c_int64(0)
        _lib.check(self._lib.tsim_profile_read_steps(self._h, C.byref(n), 1 if reset else 0), "tsim_profile_read_steps")
        return int(n.value)

    def pipeline_join(self, stream: int = 0) -> None:
        """``sample_batch_device_end`` for every slot with a launch in flight, in one call."""
        _lib.check(self._lib.tsim_pipeline_join(self._h, stream or None), "tsim_pipeline_join")

    def sample_batch_device_end(self, slot: int, stream: int = 0) -> None:
        """Make ``stream`` (0: the handle's stream) wait for the slot's second pass."""
        _lib.check(self._lib.tsim_sample_batch_device_end(self._h, int(slot), stream or None), "tsim_sample_batch_device_end")

    def pipeline_wait_slot(self, slot: int, stream: int = 0) -> None:
        """``stream`` waits for the launch ``slot`` carried last, whether or not another stream joined it already
        (``sample_batch_device_end`` is consumed by the first stream that calls it): the guard before a slot's input
        buffer is refilled."""
        _lib.check(self._lib.tsim_pipeline_wait_slot(self._h, int(slot), stream or None), "tsim_pipeline_wait_slot")

    def pipeline_lane_stream(self, lane: int) -> int:
        """``hipStream_t`` of pipeline lane ``lane`` as an integer; lane 2 is where deferred hard-row
        batches - i.e. results - complete (include/tsim_hip.h)."""
        st = C.c_void_p()
        _lib.check(self._lib.tsim_pipeline_lane_stream(self._h, int(lane), C.byref(st)), "tsim_pipeline_lane_stream")
        return int(st.value or 0)

    def pipeline_wait_stream(self, stream: int = 0) -> None:
        """Every pipeline lane waits for the work already queued on ``stream`` (0: the handle's)."""
        _lib.check(self._lib.tsim_pipeline_wait_stream(self._h, stream or None), "tsim_pipeline_wait_stream")

    def stream_ptr(self) -> int:
        """The handle's ``hipStream_t`` as an integer (to order foreign work after the kernels)."""
        st = C.c_void_p()
        _lib.check(self._lib.tsim_get_stream(self._h, C.byref(st)), "tsim_get_stream")
        return int(st.value or 0)

    def postselect_device(self, d_f: int, B: int, num_f: int, d_mask: int, d_ref: int, d_out: int,
                          d_row_index: int, d_row_count: int, d_discarded: int = 0) -> None:
        """Direct bits for every row + survivor list (asynchronous); see include/tsim_hip.h."""
        _lib.check(
            self._lib.tsim_postselect_device(
                self._h, C.c_void_p(d_f), int(B), int(num_f), C.c_void_p(d_mask),
                C.c_void_p(d_ref) if d_ref else None, C.c_void_p(d_out), C.c_void_p(d_row_index),
                C.c_void_p(d_row_count), C.c_void_p(d_discarded) if d_discarded else None, None,
            ),
            "tsim_postselect_device",
        )

    def sample_rows_device(self, d_f: int, B: int, num_f: int, key, d_out: int, d_row_index: int,
                           d_row_count: int, *, shot_offset: int = 0, d_norm_dev: int = 0) -> None:
        """Sample only the listed rows (asynchronous)."""
        _lib.check(
            self._lib.tsim_sample_rows_device(
                self._h, C.c_void_p(d_f), int(B), int(num_f), int(key[0]) & 0xFFFFFFFF,
                int(key[1]) & 0xFFFFFFFF, int(shot_offset), C.c_void_p(d_out),
                C.c_void_p(d_norm_dev) if d_norm_dev else None, C.c_void_p(d_row_index),
                C.c_void_p(d_row_count), None,
            ),
            "tsim_sample_rows_device",
        )

    def synchronize(self) -> None:
        _lib.check(self._lib.tsim_synchronize(self._h), "tsim_synchronize")

    def profile_enable(self, on=True) -> None:
        """``True``/1: time every kernel of a launch; 2: only the first kernel (cheap enough for
        pipelined launches); ``False``: off."""
        _lib.check(self._lib.tsim_profile_enable(self._h, int(on)), "tsim_profile_enable")

    def profile_set_sampling(self, every: int) -> None:
        _lib.check(self._lib.tsim_profile_set_sampling(self._h, int(every)), "tsim_profile_set_sampling")

    def profile_read(self, reset: bool = True) -> tuple[float, int]:
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self._lib.tsim_profile_read(self._h, C.byref(ms), C.byref(n), 1 if reset else 0), "tsim_profile_read")
        return float(ms.value), int(n.value)

    def profile_read_stages(self) -> dict:
        """Kernel time (ms) since the last reset, split by kernel; call before ``profile_read``."""
        st = (C.c_double * 3)()
        _lib.check(self._lib.tsim_profile_read_stages(self._h, st), "tsim_profile_read_stages")
        return {"pattern_pass": float(st[0]), "hard_rows": float(st[1]), "full_kernel": float(st[2])}

    # -- evaluate seam ----------------------------------------------------------
    def evaluate(self, component: int, level: int, param_vals: np.ndarray, *, exact: bool = False,
                 return_abs: bool = False):
        pv = np.asarray(param_vals)
        if pv.ndim != 2:
            raise ValueError(f"param_vals must be 2-D (batch, n_params), got shape {pv.shape}")
        if pv.dtype != np.uint8:
            pv = (pv != 0).astype(np.uint8)
        pv = np.ascontiguousarray(pv)
        B = pv.shape[0]
        lv = self.program.components[component].compiled_scalar_graphs[level]
        if pv.shape[1] != lv.n_params:
            raise ValueError(f"param_vals has {pv.shape[1]} columns, level expects {lv.n_params}")
        re = np.zeros(B, np.float32)
        im = np.zeros(B, np.float32)
        ex = np.zeros((B, 5), np.int32) if exact else None
        ab = np.zeros(B, np.float32) if return_abs else None
        _lib.check(
            self._lib.tsim_evaluate(self._h, component, level, _lib.ptr(pv), B, _lib.ptr(re), _lib.ptr(im),
                                    _lib.ptr(ab), _lib.ptr(ex)),
            "tsim_evaluate",
        )
        if return_abs:
            return ab  # |amplitude| formed on the device exactly as the sampling kernel forms it
        z = np.empty(B, np.complex64)
        z.real, z.imag = re, im
        return (z, ex) if exact else z


# ---------------------------------------------------------------------------
# module-level mirrors of the reference's seam functions
# ---------------------------------------------------------------------------


def get_hip_program(program, device: int = 0, mode: str = "auto", pattern_tables=None) -> HipProgram:
    """Upload ``program`` once per (device, mode, pattern_tables) and cache the handle on the
    program object."""
    if isinstance(program, HipProgram):
        return program
    cache = getattr(program, "_backend_cache", None)
    if cache is None:
        try:
            cache = {}
            object.__setattr__(program, "_backend_cache", cache)
        except Exception:  # frozen foreign object: no caching
            return HipProgram(program, device, mode, pattern_tables)
    ck = (device, mode) if pattern_tables is None else (device, mode, pattern_tables)
    hp = cache.get(ck)
    if hp is None:
        hp = HipProgram(program, device, mode, pattern_tables)
        cache[ck] = hp
    return hp


def check_norm_deviation(max_norm_deviation: float) -> None:
    """The error/warning policy of reference src/tsim/sampler.py:149-161."""
    if np.isclose(max_norm_deviation, 1):
        raise ValueError(
            "A vanishing marginal probability distribution was encountered (normalization 0). "
            "This is likely the result of an underflow error."
        )
    if max_norm_deviation > 1e-5:
        warnings.warn(
            "A marginal probability was not normalized correctly "
            f"(normalization deviated from 1 by {max_norm_deviation:.1e}). "
            "This is likely a floating point precision issue.",
            stacklevel=3,
        )


def sample_program(program, f_params, key, *, device: int = 0, mode: str = "auto") -> np.ndarray:
    """Drop-in for ``tsim.sampler.sample_program`` (sampler.py:117-167).

    ``key`` is the post-split subkey, either a ``(hi, lo)`` uint32 pair or a JAX
    typed key (``jax.random.key_data`` is applied when jax is importable).
    """
    key = _key_pair(key)
    f = np.asarray(f_params)
    if int(getattr(program, "num_outputs")) == 0:
        return np.zeros((f.shape[0], 0), dtype=np.bool_)
    hp = get_hip_program(program, device, mode)
    out, devs = hp.sample_batch(f, key)
    for dev in devs:
        check_norm_deviation(float(dev))
    return out


def evaluate(circuit: CompiledScalarGraphs, param_vals) -> np.ndarray:
    """Drop-in for ``tsim.compile.evaluate.evaluate`` (evaluate.py:15-59)."""
    from .program import CompiledComponent, make_program, scalar_graphs_from_tsim

    pv = np.asarray(param_vals)
    cache = getattr(circuit, "_eval_handle", None)
    if cache is None:
        lv = scalar_graphs_from_tsim(circuit)
        # wrap the single level as a joint-mode component: [empty norm level, this level]
        from .program import empty_scalar_graphs

        comp = CompiledComponent(
            tuple(range(lv.n_params)), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv)
        )
        prog = make_program([comp], [], lv.n_params, 0)
        cache = HipProgram(prog)
        try:
            object.__setattr__(circuit, "_eval_handle", cache)
        except Exception:
            pass
    return cache.evaluate(0, 1, pv)


def _key_pair(key) -> tuple[int, int]:
    if isinstance(key, (tuple, list)) and len(key) == 2:
        return int(key[0]), int(key[1])
    arr = None
    try:  # a JAX typed key
        import jax  # type: ignore

        arr = np.asarray(jax.random.key_data(key))
    except Exception:
        arr = np.asarray(key)
    arr = arr.reshape(-1)
    if arr.shape[0] != 2:
        raise ValueError("key must be a (hi, lo) uint32 pair or a threefry2x32 JAX key")
    return int(arr[0]), int(arr[1])


class DeviceNoiseSampler:
    """Device-side twin of ``ChannelSampler.sample`` (reference src/tsim/noise/channels.py:624-658).

    Built from a host :class:`tsim_amd.channels.ChannelSampler` (which performs the channel
    simplification once, exactly like the reference) and bound to a :class:`HipProgram`'s device and
    stream.  ``sample_into`` fills a packed ``uint64[B, ceil(num_f/64)]`` device buffer; the result
    is statistically equivalent to the host sampler, not stream-identical (numpy's PCG64 stream is
    sequential by construction).
    """

    def __init__(self, hip_program: HipProgram, channel_sampler):
        self._prog = hip_program
        self._lib = hip_program._lib
        data = channel_sampler._sparse_data
        self.num_f = int(channel_sampler.signature_matrix.shape[1])
        p_fire = np.ascontiguousarray([d[0] for d in data], dtype=np.float64)
        n_out = np.ascontiguousarray([len(d[1]) for d in data], dtype=np.int32)
        cdf = np.ascontiguousarray(np.concatenate([d[1] for d in data]) if data else np.zeros(0), dtype=np.float64)
        pats = (
            np.ascontiguousarray(np.concatenate([d[2] for d in data], axis=0), dtype=np.uint8)
            if data else np.zeros((0, self.num_f), np.uint8)
        )
        h = C.c_void_p()
        _lib.check(
            self._lib.tsim_noise_create(
                hip_program._h, self.num_f, len(data), _lib.ptr(p_fire), _lib.ptr(n_out), _lib.ptr(cdf),
                _lib.ptr(pats), C.byref(h),
            ),
            "tsim_noise_create",
        )
        self._n = h
        self._handle = hip_program._handle
        self._handle.noise.append(h)
        self._finalizer = weakref.finalize(self, DeviceNoiseSampler._destroy, self._handle, h)

    @staticmethod
    def _destroy(handle: _Handle, n) -> None:
        if handle.h and n in handle.noise:  # otherwise _Handle.close already destroyed it
            handle.noise.remove(n)
            handle.lib.tsim_noise_destroy(n)

    def sample_into(self, d_f: int, B: int, key, stream: int = 0) -> None:
        """Asynchronous on ``stream`` (0: the program's stream)."""
        if not self._handle.h:
            raise _lib.HipBackendError("the HipProgram of this noise sampler is closed")
        _lib.check(
            self._lib.tsim_noise_sample_device(
                self._n, int(B), int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF, C.c_void_p(d_f), stream or None
            ),
            "tsim_noise_sample_device",
        )

    def sample(self, B: int, key) -> np.ndarray:
        """Convenience: ``uint8[B, num_f]`` on the host (for tests)."""
        wf = max(1, (self.num_f + 63) // 64)
        buf = self._prog.malloc(max(1, B) * wf * 8)
        self.sample_into(buf.ptr, B, key)
        packed = np.zeros((B, wf * 8), np.uint8)
        self._prog.d2h(packed, buf)
        buf.free()
        return np.unpackbits(packed, axis=1, bitorder="little")[:, : self.num_f]


class _PinnedBuf:
    """Owns one ``hipHostMalloc`` allocation (freed when the last numpy view dies)."""

    def __init__(self, nbytes: int):
        self._lib = _lib.load()
        p = C.c_void_p()
        _lib.check(self._lib.tsim_malloc_pinned(int(nbytes), C.byref(p)), "tsim_malloc_pinned")
        self.ptr = int(p.value or 0)
        self.nbytes = int(nbytes)

    def __del__(self):  # pragma: no cover - interpreter teardown order is not guaranteed
        try:
            if self.ptr:
                self._lib.tsim_free_pinned(C.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass


class PinnedPool:
    """Result arrays on recycled pinned host memory.

    A fresh ``np.empty`` of tens of MB costs more than the PCIe copy that fills it (the kernel zeroes every page on first
    touch: ~5 ms per 48 MB, where the copy takes 0.9 ms), and so does a fresh ``hipHostMalloc``.  ``take`` hands out a
    numpy array backed by a pinned block; when the LAST view of it dies the block comes back to the pool (a
    ``weakref.finalize`` on the exporting buffer) instead of being unpinned, so the next call of a sampling loop gets
    warm, pinned memory: asynchronous copies at the full PCIe rate, no page faults.  Blocks the caller keeps alive are
    simply not reused; at most ``cap`` bytes wait in the pool."""

    def __init__(self, cap: int = 2 << 30):
        self.cap = int(cap)
        self._free: list[_PinnedBuf] = []

    def _give_back(self, buf: "_PinnedBuf") -> None:
        if sum(b.nbytes for b in self._free) + buf.nbytes <= self.cap:
            self._free.append(buf)
        # else: dropped here -> _PinnedBuf.__del__ unpins it

    def take(self, shape, dtype=np.uint8) -> np.ndarray:
        import weakref

        need = max(1, int(np.prod(shape)) * np.dtype(dtype).itemsize)
        best = None
        for i, b in enumerate(self._free):
            if need <= b.nbytes <= max(2 * need, need + (1 << 20)) and (best is None or b.nbytes < self._free[best].nbytes):
                best = i
        buf = self._free.pop(best) if best is not None else _PinnedBuf(need)
        carr = (C.c_uint8 * buf.nbytes).from_address(buf.ptr)
        weakref.finalize(carr, self._give_back, buf)
        return np.frombuffer(carr, dtype=np.uint8)[:need].view(dtype).reshape(shape)


_result_pool = PinnedPool()


def result_pool() -> PinnedPool:
    return _result_pool


def alloc_pinned_numpy(nbytes: int, dtype, shape) -> np.ndarray:
    """Pinned host ndarray (mirror of reference src/tsim/utils/cuda_helpers.py:73-102, on hipHostMalloc)."""
    buf = _PinnedBuf(max(1, int(nbytes)))
    carr = (C.c_uint8 * buf.nbytes).from_address(buf.ptr)
    carr._owner = buf  # keeps the allocation alive as long as any view exists
    return np.frombuffer(carr, dtype=np.uint8)[: int(np.prod(shape)) * np.dtype(dtype).itemsize].view(dtype).reshape(shape)
