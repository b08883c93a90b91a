"""Compiled samplers: host orchestration around the fused HIP kernel.

Same classes, keyword surface and semantics as the reference's
``src/tsim/sampler.py:170-953`` (``CompiledMeasurementSampler``, ``CompiledDetectorSampler``,
``CompiledStateProbs``), minus the circuit front-end: a sampler is built from an already
compiled program plus the noise model (``channel_probs``, ``error_transform``), e.g. loaded from
an ``.npz`` exported where tsim is installed (``from_npz``).

What stays identical for a fixed ``(seed, batch_size)``:

* the host key is split once per batch and the subkey goes to ``sample_program``
  (reference sampler.py:399), once more for a reference sample (:272) and once per dispatched
  survivor batch under post-selection (:482);
* the channel sampler is seeded with ``default_rng(seed).integers(0, 2**30)`` (:203) and asked for
  exactly ``batch_size`` rows per batch (:393);
* reference-sample handling (batch bump :384-387, row 0 zeroed :395-396, stripped :402-404),
  the direct fast path (:547-555), post-selection buffering/padding (:422-545) and the output
  column arrangement / bit packing (:850-868).

``sample_program`` is looked up as a module attribute at call time, exactly like the reference, so
tests can spy on it.
"""

from __future__ import annotations

from math import ceil

import numpy as np

from . import prng
from .backend import (  # noqa: F401
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


class _CompiledSamplerBase:
    """Shared state: key, compiled program, channel sampler, direct-output tables."""

    _PIPELINE = 8  # launches in flight on the device path (one f buffer each)

    def __init__(
        self,
        program,
        *,
        channel_probs: list,
        error_transform: np.ndarray,
        seed: int | None = None,
        device: int = 0,
        noise: str = "host",
        mode: str = "auto",
    ):
        """``noise="host"`` reproduces the reference's numpy channel stream bit for bit;
        ``noise="device"`` samples the channels on the GPU (statistically equivalent, the error
        bits never leave HBM) - the throughput mode of SURVEY.md section 7, hard part 3."""
        if noise not in ("host", "device"):
            raise ValueError("noise must be 'host' or 'device'")
        self._noise = noise
        self._mode = mode  # kernel formulation: "auto" | "rows" | "faithful" (see HipProgram)
        self._noise_key = None
        self._device_state = None
        if seed is None:
            seed = int(np.random.default_rng().integers(0, 2**30))
        self._key = prng.key(seed)
        self._program: CompiledProgram = from_tsim(program)
        self._device = int(device)
        channel_seed = int(np.random.default_rng(seed).integers(0, 2**30))
        self._channel_sampler = ChannelSampler(
            channel_probs=channel_probs, error_transform=error_transform, seed=channel_seed
        )
        self._num_detectors = int(self._program.num_detectors)
        self._noise_key = prng.key(channel_seed)  # key chain of the device noise sampler

        prog = self._program
        self._direct_f_indices = np.asarray(prog.direct_f_indices)
        self._direct_flips = np.asarray(prog.direct_flips, dtype=np.bool_)
        self._direct_reindex = None if prog.output_reindex is None else np.asarray(prog.output_reindex)
        n_direct = len(self._direct_f_indices)
        # zero-copy case: f indices 0..n-1, no flips, no reindex (typical surface-code detectors)
        self._direct_zero_copy = (
            n_direct > 0
            and self._direct_reindex is None
            and not self._direct_flips.any()
            and np.array_equal(self._direct_f_indices, np.arange(n_direct))
        )
        self._direct_global_indices = np.asarray(prog.output_order[:n_direct], dtype=np.int32)
        self._direct_output_mask = np.zeros(prog.num_outputs, dtype=np.bool_)
        if n_direct > 0:
            self._direct_output_mask[self._direct_global_indices] = True
        self._direct_detector_mask = self._direct_output_mask[: self._num_detectors].copy()

    # -- construction from an exported program ---------------------------------
    @classmethod
    def from_npz(cls, path, *, seed: int | None = None, device: int = 0, **kw):
        """Load a program + noise model written by ``program.save_npz(path, prog, channel_probs_k=..., error_transform=...)``."""
        prog, extra = load_npz(path)
        n = int(extra["n_channels"])
        probs = [np.asarray(extra[f"channel_probs_{i}"], dtype=np.float64) for i in range(n)]
        return cls(prog, channel_probs=probs, error_transform=extra["error_transform"], seed=seed, device=device, **kw)

    # -- direct outputs ------------------------------------------------------------
    def _compute_direct_outputs(self, f_params_np: np.ndarray) -> np.ndarray:
        """Direct bits scattered into a full ``(batch, num_outputs)`` bool array (others False)."""
        batch = f_params_np.shape[0]
        num_outputs = self._program.num_outputs
        n_direct = len(self._direct_f_indices)
        if n_direct == 0:
            return np.zeros((batch, num_outputs), dtype=np.bool_)
        if self._direct_zero_copy:
            raw = f_params_np[:, :n_direct].view(np.bool_)
            if n_direct == num_outputs:
                return raw.copy()
        else:
            raw = (f_params_np[:, self._direct_f_indices] ^ self._direct_flips).view(np.bool_)
        out = np.zeros((batch, num_outputs), dtype=np.bool_)
        out[:, self._direct_global_indices] = raw
        return out

    def _compute_reference_sample(self) -> np.ndarray:
        """Noiseless sample (all f = 0); consumes one key split iff there are compiled components."""
        num_f = self._channel_sampler.signature_matrix.shape[1]
        f_ref = np.zeros((1, num_f), dtype=np.uint8)
        if not self._program.components:
            return self._compute_direct_outputs(f_ref)[0]
        self._key, subkey = prng.split(self._key)
        return np.asarray(_call_sample_program(self, f_ref, subkey)[0], dtype=np.bool_)

    # -- batch sizing ----------------------------------------------------------------
    def _peak_bytes_per_sample(self) -> int:
        """Device bytes per shot of the fused path: unpacked f + packed f + packed/unpacked outputs.

        (The reference's estimate counts its materialised [B,G,T,4] tensors, sampler.py:294-306;
        the fused kernel keeps all of that in registers.)
        """
        num_f = int(self._channel_sampler.signature_matrix.shape[1])
        n_out = int(self._program.num_outputs)
        return max(1, num_f + 8 * ((num_f + 63) // 64) + n_out + 8 * ((n_out + 63) // 64))

    def _estimate_batch_size(self) -> int:
        """Largest batch worth launching: a quarter of HBM, capped at 2**24 shots."""
        available = 64 * 1024**3  # conservative share of the 288 GB of one MI355X
        return max(1, min(1 << 24, int(available * 0.25) // self._peak_bytes_per_sample()))

    def _resolve_batch_size(self, shots: int, batch_size: int | None, *, compute_reference: bool) -> int:
        if batch_size is None:
            max_batch_size = self._estimate_batch_size()
            num_batches = max(1, ceil(shots / max_batch_size))
            batch_size = ceil(shots / num_batches)
        if compute_reference and batch_size * ceil(shots / batch_size) == shots:
            batch_size += 1
        return batch_size

    # -- plain batching ----------------------------------------------------------------
    def _sample_batches(self, shots: int, batch_size: int | None = None, *, compute_reference: bool = False):
        if shots < 0:
            raise ValueError(f"shots must be non-negative, got {shots}")
        if batch_size is not None and batch_size < 1:
            raise ValueError(f"batch_size must be at least 1, got {batch_size}")
        num_outputs = self._program.num_outputs
        if shots == 0:
            empty = np.empty((0, num_outputs), dtype=np.bool_)
            return (empty, np.zeros(num_outputs, dtype=np.bool_)) if compute_reference else empty
        if not self._program.components:
            samples = self._sample_direct(shots)
            return (samples, self._compute_reference_sample()) if compute_reference else samples
        if self._noise == "device":
            return self._sample_batches_device(shots, batch_size, compute_reference=compute_reference)

        if batch_size is None:
            max_batch_size = self._estimate_batch_size()
            num_batches = max(1, ceil(shots / max_batch_size))
            batch_size = ceil(shots / num_batches)
        else:
            num_batches = ceil(shots / batch_size)
        if compute_reference and batch_size * num_batches == shots:
            batch_size += 1  # room for the reference row, uniform batch shapes kept

        batches = []
        reference = None
        # With the library's own sample_program in place (not a test replacement) the error rows go to the
        # GPU packed: same generator stream, same bits, 8x fewer bytes to scatter and to copy.
        packed_route = globals()["sample_program"] is _backend_sample_program
        for _ in range(num_batches):
            want_ref = compute_reference and reference is None
            if packed_route:
                f_packed = self._channel_sampler.sample_packed(batch_size)
                if want_ref:
                    f_packed[0] = 0
                self._key, subkey = prng.split(self._key)
                hp = get_hip_program(self._program, self._device, self._mode)
                num_f = int(self._channel_sampler.signature_matrix.shape[1])
                samples, devs = hp.sample_batch_packed(f_packed, num_f, subkey)
                for dev in devs:
                    check_norm_deviation(float(dev))
            else:
                f_params_np = self._channel_sampler.sample(batch_size)
                if want_ref:
                    f_params_np[0] = 0
                self._key, subkey = prng.split(self._key)
                samples = _call_sample_program(self, f_params_np, subkey)
            if want_ref:
                reference = np.asarray(samples[0])
                samples = samples[1:]
            batches.append(samples)
        result = (batches[0] if len(batches) == 1 else np.concatenate(batches, axis=0))[:shots]
        if compute_reference:
            assert reference is not None
            return result, reference
        return result

    # -- device-resident pipeline (noise="device") ---------------------------------------------
    def _sample_batches_device(self, shots: int, batch_size: int | None, *, compute_reference: bool = False,
                               packed_columns: int | None = None):
        """noise -> f -> sample_program -> unpack, all on the GPU; one pinned D2H at the end.

        Per batch: one split of the noise key chain for the device channel sampler and one split
        of the sampler key for ``sample_program`` (as reference sampler.py:399).  The reference
        sample, when requested, is its own 1-row call with f = 0 (sampler.py:263-276).
        """
        hp = get_hip_program(self._program, self._device, self._mode)
        st = self._device_state
        if st is None or st["hp"] is not hp:
            st = self._device_state = dict(hp=hp, noise=DeviceNoiseSampler(hp, self._channel_sampler), bufs=None)
        if batch_size is None:
            max_batch_size = self._estimate_batch_size()
            num_batches = max(1, ceil(shots / max_batch_size))
            batch_size = ceil(shots / num_batches)
        else:
            num_batches = ceil(shots / batch_size)
        reference = self._compute_reference_sample() if compute_reference else None
        num_f = int(self._channel_sampler.signature_matrix.shape[1])
        n_out = int(self._program.num_outputs)
        wf, wo = max(1, (num_f + 63) // 64), (n_out + 63) // 64
        total = num_batches * batch_size
        n_comp = max(1, len(self._program.components))
        need = (batch_size, total)
        if st["bufs"] is None or st["bufs"]["need"] != need:
            st["bufs"] = dict(
                need=need,
                f=[hp.malloc(batch_size * wf * 8) for _ in range(self._PIPELINE)],
                out=hp.malloc(total * wo * 8),
                u8=hp.malloc(total * n_out),
                devs=hp.malloc(num_batches * n_comp * 4),
            )
        b = st["bufs"]
        # pipelined launches (include/tsim_hip.h: tsim_sample_batch_device_begin/_end): batch i's
        # hard-row pass overlaps the noise sampling and first pass of batch i+1 (each slot is a lane);
        # each slot owns an f buffer, every batch its own slice of the output buffer
        for i in range(num_batches):
            slot = i % self._PIPELINE
            hp.sample_batch_device_end(slot)  # the slot's previous batch no longer reads its f buffer
            self._noise_key, nk = hp.split_key(self._noise_key)  # prng.split, computed by the library
            st["noise"].sample_into(b["f"][slot].ptr, batch_size, nk)
            self._key, subkey = hp.split_key(self._key)
            hp.sample_batch_device_begin(
                slot, b["f"][slot].ptr, batch_size, num_f, subkey, b["out"].ptr + i * batch_size * wo * 8,
                d_norm_dev=b["devs"].ptr + i * n_comp * 4,
            )
        for slot in range(self._PIPELINE):
            hp.sample_batch_device_end(slot)
        if packed_columns is not None:
            # bit_packed=True of the first `packed_columns` columns: ceil(n/8) bytes per shot over PCIe
            rb = (packed_columns + 7) // 8
            hp.compact_rows_device(b["out"].ptr, total, packed_columns, b["u8"].ptr, in_words=wo)
            result = np.empty((total, rb), np.uint8)
            hp.d2h(result, b["u8"].ptr)
            devs = np.zeros(num_batches * n_comp, np.float32)
            hp.d2h(devs, b["devs"])
            for dev in devs[: num_batches * len(self._program.components)]:
                check_norm_deviation(float(dev))
            return result[:shots]
        hp.unpack_bits_device(b["out"].ptr, total, n_out, b["u8"].ptr)
        # straight into a pageable array: a fresh pinned allocation of this size costs more (19 ms
        # per 80 MB, hipHostMalloc) than the whole pipeline; the copy itself runs at ~35 GB/s
        result = np.empty((total, n_out), np.uint8)
        hp.d2h(result, b["u8"])
        devs = np.zeros(num_batches * n_comp, np.float32)
        hp.d2h(devs, b["devs"])
        for dev in devs[: num_batches * len(self._program.components)]:
            check_norm_deviation(float(dev))
        result = result.view(np.bool_)[:shots]
        return (result, reference) if compute_reference else result

    def _postselect_device(self, shots, batch_size, *, postselect_direct, compute_reference, xor_detector_ref):
        """Post-selection with everything on the GPU: ``k_noise`` -> ``k_direct_filter`` (direct bits
        for all rows + survivor list) -> the sampling kernel on the survivors only.  Same return
        contract as the host path: ``(result, reference, was_discarded)``."""
        hp = get_hip_program(self._program, self._device, self._mode)
        st = self._device_state
        if st is None or st["hp"] is not hp:
            st = self._device_state = dict(hp=hp, noise=DeviceNoiseSampler(hp, self._channel_sampler), bufs=None)
        if batch_size is None:
            batch_size = self._resolve_batch_size(shots, batch_size, compute_reference=False)
        num_batches = ceil(shots / batch_size)
        reference = self._compute_reference_sample() if compute_reference else None
        nd = self._num_detectors
        num_f = int(self._channel_sampler.signature_matrix.shape[1])
        n_out = int(self._program.num_outputs)
        wf, wo = max(1, (num_f + 63) // 64), (n_out + 63) // 64
        total = num_batches * batch_size
        n_comp = max(1, len(self._program.components))

        def pack_cols(bits_nd):
            full = np.zeros(wo * 64, np.uint8)
            full[:nd] = np.asarray(bits_nd, np.uint8)
            return np.packbits(full, bitorder="little").view(np.uint64)

        mask_w = pack_cols(postselect_direct)
        use_ref = xor_detector_ref and reference is not None
        ref_w = pack_cols(reference[:nd]) if use_ref else None
        d_f, d_out = hp.malloc(batch_size * wf * 8), hp.malloc(total * wo * 8)
        d_u8, d_disc = hp.malloc(total * n_out), hp.malloc(total)
        d_idx, d_cnt = hp.malloc(batch_size * 4), hp.malloc(4)
        d_devs, d_mask = hp.malloc(num_batches * n_comp * 4), hp.malloc(wo * 8)
        d_ref = hp.malloc(wo * 8)
        hp.h2d(d_mask, mask_w)
        if use_ref:
            hp.h2d(d_ref, ref_w)
        hp.h2d(d_devs, np.zeros(num_batches * n_comp, np.float32))
        for i in range(num_batches):
            self._noise_key, nk = hp.split_key(self._noise_key)
            st["noise"].sample_into(d_f.ptr, batch_size, nk)
            out_i = d_out.ptr + i * batch_size * wo * 8
            hp.postselect_device(d_f.ptr, batch_size, num_f, d_mask.ptr, d_ref.ptr if use_ref else 0, out_i,
                                 d_idx.ptr, d_cnt.ptr, d_disc.ptr + i * batch_size)
            self._key, subkey = hp.split_key(self._key)
            hp.sample_rows_device(d_f.ptr, batch_size, num_f, subkey, out_i, d_idx.ptr, d_cnt.ptr,
                                  d_norm_dev=d_devs.ptr + i * n_comp * 4)
        hp.unpack_bits_device(d_out.ptr, total, n_out, d_u8.ptr)
        result = np.empty((total, n_out), np.uint8)  # pageable on purpose, see _sample_batches_device
        hp.d2h(result, d_u8)
        disc = np.zeros(total, np.uint8)
        hp.d2h(disc, d_disc)
        devs = np.zeros(num_batches * n_comp, np.float32)
        hp.d2h(devs, d_devs)
        for dev in devs[: num_batches * len(self._program.components)]:
            check_norm_deviation(float(dev))
        for buf in (d_f, d_out, d_u8, d_disc, d_idx, d_cnt, d_devs, d_mask, d_ref):
            buf.free()
        result = result.view(np.bool_)[:shots]
        was_discarded = disc[:shots].astype(np.bool_)
        # discarded rows keep their direct DETECTOR columns only (reference sampler.py:519-521)
        result[was_discarded, nd:] = False
        if use_ref:
            det_ref = reference[:nd]
            result[~was_discarded, :nd] ^= det_ref
            result[was_discarded, :nd] ^= det_ref & self._direct_detector_mask
        return result, reference, was_discarded

    # -- post-selection ------------------------------------------------------------------
    def _sample_batches_with_postselection(
        self,
        shots: int,
        batch_size: int | None,
        *,
        postselection_mask: np.ndarray,
        compute_reference: bool = False,
        xor_detector_ref: bool = False,
    ):
        """Shots discarded by a masked *direct* detector never reach the device (sampler.py:422-545)."""
        if shots < 0:
            raise ValueError(f"shots must be non-negative, got {shots}")
        if batch_size is not None and batch_size < 1:
            raise ValueError(f"batch_size must be at least 1, got {batch_size}")
        num_outputs = self._program.num_outputs
        nd = self._num_detectors
        if shots == 0:
            empty = np.empty((0, num_outputs), dtype=np.bool_)
            none_discarded = np.empty(0, dtype=np.bool_)
            ref0 = np.zeros(num_outputs, dtype=np.bool_) if compute_reference else None
            return empty, ref0, none_discarded

        postselect_direct = postselection_mask & self._direct_detector_mask
        if self._noise == "device" and self._program.components:
            return self._postselect_device(
                shots, batch_size, postselect_direct=postselect_direct,
                compute_reference=compute_reference, xor_detector_ref=xor_detector_ref,
            )
        if not self._program.components:
            samples = self._sample_direct(shots)
            reference = None
            if compute_reference:
                reference = self._compute_reference_sample()
                if xor_detector_ref:
                    samples[:, :nd] ^= reference[:nd]
            return samples, reference, np.zeros(shots, dtype=np.bool_)

        if batch_size is None:
            batch_size = self._resolve_batch_size(shots, batch_size, compute_reference=False)
        reference = self._compute_reference_sample() if compute_reference else None

        result = np.zeros((shots, num_outputs), dtype=np.bool_)
        was_discarded = np.zeros(shots, dtype=np.bool_)
        pending_f: list = []  # survivor f rows not yet dispatched
        pending_idx: list = []

        def dispatch(f_batch: np.ndarray, indices: list, n_valid: int) -> None:
            self._key, subkey = prng.split(self._key)
            out = np.asarray(_call_sample_program(self, f_batch, subkey))
            result[indices[:n_valid]] = out[:n_valid]

        def flush(final: bool = False) -> None:
            nonlocal pending_f, pending_idx
            while len(pending_f) >= batch_size:
                dispatch(np.stack(pending_f[:batch_size]), pending_idx[:batch_size], batch_size)
                pending_f, pending_idx = pending_f[batch_size:], pending_idx[batch_size:]
            if final and pending_f:
                n_valid = len(pending_f)
                stack = np.stack(pending_f)
                f_batch = np.empty((batch_size, stack.shape[1]), dtype=stack.dtype)
                f_batch[:n_valid] = stack
                f_batch[n_valid:] = stack[0]  # padding keeps the batch shape fixed
                dispatch(f_batch, pending_idx, n_valid)
                pending_f, pending_idx = [], []

        done = 0
        while done < shots:
            chunk = min(batch_size, shots - done)
            f_params_np = self._channel_sampler.sample(chunk)
            direct_full = self._compute_direct_outputs(f_params_np)
            det_cols = direct_full[:, :nd]
            if xor_detector_ref and reference is not None:
                det_cols = det_cols ^ reference[:nd]
            discarded = (det_cols & postselect_direct).any(axis=1)
            result[done : done + chunk, :nd] = direct_full[:, :nd]
            was_discarded[done : done + chunk] = discarded
            keep = np.flatnonzero(~discarded)
            if keep.size:
                pending_f.extend(f_params_np[keep])
                pending_idx.extend((done + keep).tolist())
            done += chunk
            flush()
        flush(final=True)

        if xor_detector_ref and reference is not None:
            det_ref = reference[:nd]
            result[~was_discarded, :nd] ^= det_ref
            result[was_discarded, :nd] ^= det_ref & self._direct_detector_mask
        return result, reference, was_discarded

    def _sample_direct(self, shots: int) -> np.ndarray:
        """All outputs direct: pure numpy, the device is never touched (sampler.py:547-555)."""
        f_params = self._channel_sampler.sample(shots)
        if self._direct_zero_copy:
            return f_params[:, : len(self._direct_f_indices)].view(np.bool_)
        result = f_params[:, self._direct_f_indices] ^ self._direct_flips
        if self._direct_reindex is not None:
            result = result[:, self._direct_reindex]
        return result.view(np.bool_)

    def __repr__(self) -> str:
        """Compilation statistics in the reference's format (sampler.py:557-609)."""
        graphs, params, na, nb, nc, nd_, outs = [], [], [], [], [], [], []
        nbytes = 0
        for comp in self._program.components:
            for lv in comp.compiled_scalar_graphs:
                outs.append(len(comp.output_indices))
                graphs.append(lv.num_graphs)
                params.append(lv.n_params)
                na.append(np.asarray(lv.node_phases.phases).size)
                nb.append(np.asarray(lv.halfpi_phases.coeffs).size)
                nc.append(np.asarray(lv.pi_products.psi_const).size)
                nd_.append(np.asarray(lv.phase_pairs.alpha).size + np.asarray(lv.phase_pairs.beta).size)
                for fam in (lv.node_phases, lv.halfpi_phases, lv.pi_products, lv.phase_pairs, lv.prefactor):
                    nbytes += sum(np.asarray(v).nbytes for v in vars(fam).values() if isinstance(v, np.ndarray))

        def fmt(n: int) -> str:
            if n < 1024:
                return f"{n} B"
            return f"{n / 1024:.1f} kB" if n < 1024**2 else f"{n / 1024**2:.1f} MB"

        bits = sum(c.num_bits for c in self._channel_sampler.channels)
        return (
            f"{type(self).__name__}({len(self._program.direct_f_indices)} direct, "
            f"{int(np.sum(graphs))} graphs, {bits} error channel bits, "
            f"{max(outs) if outs else 0} outputs for largest cc, "
            f"≤ {max(params) if params else 0} parameters, {int(np.sum(na))} A terms, "
            f"{int(np.sum(nb))} B terms, {int(np.sum(nc))} C terms, {int(np.sum(nd_))} D terms, "
            f"{fmt(nbytes)})"
        )


def _call_sample_program(sampler: _CompiledSamplerBase, f_params: np.ndarray, subkey) -> np.ndarray:
    """Late-bound call of this module's ``sample_program`` (so tests can replace it)."""
    fn = globals()["sample_program"]
    if fn is _backend_sample_program:
        return fn(sampler._program, f_params, subkey, device=sampler._device, mode=sampler._mode)
    return fn(sampler._program, f_params, subkey)  # a replacement with the reference's 3-arg signature


class CompiledMeasurementSampler(_CompiledSamplerBase):
    """Samples measurement outcomes (sequential levels 0..n per component)."""

    def sample(self, shots: int, *, batch_size: int | None = None) -> np.ndarray:
        return self._sample_batches(shots, batch_size)


def _maybe_bit_pack(array: np.ndarray, *, bit_packed: bool) -> np.ndarray:
    if not bit_packed:
        return array
    return np.packbits(array.astype(np.bool_), axis=1, bitorder="little")


class CompiledDetectorSampler(_CompiledSamplerBase):
    """Samples detector and observable outcomes."""

    def sample(
        self,
        shots: int,
        *,
        batch_size: int | None = None,
        prepend_observables: bool = False,
        append_observables: bool = False,
        separate_observables: bool = False,
        bit_packed: bool = False,
        use_detector_reference_sample: bool = False,
        use_observable_reference_sample: bool = False,
        postselection_mask: np.ndarray | None = None,
    ):
        """Detector samples, with the reference's column-arrangement flags (sampler.py:732-868)."""
        if separate_observables and (prepend_observables or append_observables):
            raise ValueError(
                "Can't specify separate_observables=True with append_observables=True or prepend_observables=True"
            )
        nd = self._num_detectors
        compute_reference = use_detector_reference_sample or use_observable_reference_sample

        if postselection_mask is not None:
            mask = np.asarray(postselection_mask, dtype=np.bool_)
            if mask.shape != (nd,):
                raise ValueError(f"postselection_mask must have shape ({nd},), got {mask.shape}")
            postselection_mask = mask
            if not (mask & self._direct_detector_mask).any() or not self._program.components:
                postselection_mask = None  # nothing can be skipped: plain path

        if postselection_mask is not None:
            samples, reference, discarded = self._sample_batches_with_postselection(
                shots,
                batch_size,
                postselection_mask=postselection_mask,
                compute_reference=compute_reference,
                xor_detector_ref=use_detector_reference_sample,
            )
            if compute_reference and use_observable_reference_sample:
                samples[~discarded, nd:] ^= reference[nd:]
        elif compute_reference:
            samples, reference = self._sample_batches(shots, batch_size, compute_reference=True)
            if use_detector_reference_sample:
                samples[:, :nd] ^= reference[:nd]
            if use_observable_reference_sample:
                samples[:, nd:] ^= reference[nd:]
        else:
            ncols = self._program.num_outputs if append_observables else nd
            if (bit_packed and self._noise == "device" and not separate_observables and not prepend_observables
                    and ncols > 0 and shots > 0 and self._program.components
                    and (batch_size is None or batch_size >= 1)):
                # the requested columns are the first `ncols` bits of a packed device row: compact on
                # the GPU and move ceil(ncols/8) bytes per shot instead of one byte per bit
                return self._sample_batches_device(shots, batch_size, packed_columns=ncols)
            samples = self._sample_batches(shots, batch_size)

        det, obs = samples[:, :nd], samples[:, nd:]
        if separate_observables:
            return _maybe_bit_pack(det, bit_packed=bit_packed), _maybe_bit_pack(obs, bit_packed=bit_packed)
        if prepend_observables and append_observables:
            cols = np.concatenate([obs, det, obs], axis=1)
        elif append_observables:
            cols = samples
        elif prepend_observables:
            cols = np.concatenate([obs, det], axis=1)
        else:
            cols = det
        return _maybe_bit_pack(cols, bit_packed=bit_packed)


class CompiledStateProbs(_CompiledSamplerBase):
    """``P(state | error sample)`` from joint-mode programs (levels [0, n] per component)."""

    def probability_of(self, state: np.ndarray, *, batch_size: int) -> np.ndarray:
        """sampler.py:906-953: ``p_joint / p_norm`` per sampled error configuration."""
        if batch_size < 1:
            raise ValueError(f"batch_size must be at least 1, got {batch_size}")
        state = np.asarray(state)
        expected = self._program.num_outputs
        if state.shape != (expected,):
            raise ValueError(f"state must have shape ({expected},), got {state.shape}")
        f_samples = self._channel_sampler.sample(batch_size)
        p_norm = np.ones(batch_size, dtype=np.float32)
        p_joint = np.ones(batch_size, dtype=np.float32)
        n_direct = len(self._program.direct_f_indices)
        if n_direct > 0:
            direct_bits = f_samples[:, self._direct_f_indices].astype(np.bool_) ^ self._direct_flips
            targets = state[np.asarray(self._program.output_order[:n_direct])].astype(np.bool_)
            p_joint = p_joint * (direct_bits == targets).all(axis=1).astype(np.float32)
        hp = get_hip_program(self._program, self._device, self._mode)
        for ci, comp in enumerate(self._program.components):
            assert len(comp.compiled_scalar_graphs) == 2
            f_sel = f_samples[:, np.asarray(comp.f_selection, dtype=np.int64)]
            p_norm = p_norm * hp.evaluate(ci, 0, f_sel, return_abs=True)
            comp_state = state[list(comp.output_indices)].astype(np.uint8)
            joint = np.hstack([f_sel, np.tile(comp_state, (batch_size, 1))])
            p_joint = p_joint * hp.evaluate(ci, 1, joint, return_abs=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.asarray(p_joint / p_norm)
