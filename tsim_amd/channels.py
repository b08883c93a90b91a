"""Noise model -> error-mechanism rows ``f`` (host side of the path).

What must come out is fixed by the reference (``src/tsim/noise/channels.py``): for a seed, the same
simplified channel tables (float64, bit for bit) and the same ``uint8[num_samples, num_f]`` array as its
``ChannelSampler`` - that is the drop-in contract, pinned by golden vectors generated from the reference
module itself (``tests/golden/gen_channels_golden.py``).  HOW it is computed here is this repo's own:

* **Signatures as packed bit strings.**  ``f_i = XOR_j error_transform[i, j] e_j``: an error bit matters only
  through its column of ``error_transform``.  Columns are packed into integers (row 0 most significant, so
  integer order is the lexicographic column order the reference's ``np.unique(axis=1)`` yields) and
  numbered in sorted order; the XOR pattern of a channel outcome is an XOR of packed signature words -
  produced directly in the ``uint64`` little-endian row layout the sampling kernels read.
* **One structural primitive.**  Dropping a bit that touches no ``f``, sorting a channel's bits, folding
  bits with the same signature and embedding a channel into a wider one are all the same operation: the
  push-forward of an outcome distribution along a GF(2)-linear map of the outcome bits
  (:func:`push_forward`).  Channels on nested supports are then XOR-convolved (:func:`combine_supports`).
* **Packed rows are the product.**  :meth:`ChannelSampler.sample_packed` scatters 8-byte words;
  :meth:`ChannelSampler.sample` is its unpacked view for callers of the reference layout.

Forced by the stream contract (and only that): the order of float additions inside
:func:`push_forward`/:func:`xor_convolve`/:func:`marginalise`, and the generator-call sequence of
:meth:`ChannelSampler._draw` (``Generator.geometric`` then ``Generator.uniform`` per channel with the
reference's draw-count formula, ``channels.py:624-658``) together with the float sequence that builds the
conditional CDFs (``channels.py:578-622``).
"""

from __future__ import annotations

import ctypes as C
import warnings
from dataclasses import dataclass

import numpy as np

_TOL = 1e-6


@dataclass
class Channel:
    """A distribution over the ``2**k`` joint values of ``k`` error bits.

    ``probs[o]``: probability that the bits read ``o`` (bit ``i`` of ``o`` = error bit ``i``);
    ``unique_col_ids[i]``: number of the ``error_transform`` column signature that bit ``i`` carries.
    """

    probs: np.ndarray
    unique_col_ids: tuple

    def __post_init__(self) -> None:
        p = np.asarray(self.probs, dtype=np.float64)
        if p.ndim != 1 or p.size != 1 << len(self.unique_col_ids):
            raise ValueError(f"a channel over {len(self.unique_col_ids)} bits needs {1 << len(self.unique_col_ids)} probabilities, got shape {p.shape}")
        if p.min(initial=0.0) < -_TOL or p.max(initial=0.0) > 1.0 + _TOL:
            raise ValueError(f"channel probabilities outside [0, 1]: {p}")
        total = float(p.sum())
        if not np.isclose(total, 1.0):
            raise ValueError(f"channel probabilities add up to {total}, not 1: {p}")

    @property
    def num_bits(self) -> int:
        return len(self.unique_col_ids)


# ---- outcome tables of the noise instructions (bit orders are the reference's, channels.py:47-198) -------


def _outcome_table(num_bits: int, fires: dict, stay: float) -> np.ndarray:
    """Outcome distribution of a noise instruction: ``fires[outcome]`` for the listed non-identity outcomes,
    ``stay`` for outcome 0, zero elsewhere (bit ``i`` of an outcome = the instruction's error bit ``i``)."""
    table = np.zeros(1 << num_bits, dtype=np.float64)
    table[0] = stay
    for outcome, prob in fires.items():
        table[outcome] = prob
    return table


def error_probs(p: float) -> np.ndarray:
    """One error bit that fires with probability ``p``."""
    return _outcome_table(1, {0b1: p}, 1 - p)


def pauli_channel_1_probs(px: float, py: float, pz: float) -> np.ndarray:
    """Single-qubit Pauli channel; bit 0 = Z component, bit 1 = X component (Y = both)."""
    return _outcome_table(2, {0b01: pz, 0b10: px, 0b11: py}, 1 - px - py - pz)


def heralded_pauli_channel_1_probs(pi: float, px: float, py: float, pz: float) -> np.ndarray:
    """Heralded Pauli channel; bit 0 = herald, bit 1 = Z, bit 2 = X.  Nothing fires without the herald."""
    return _outcome_table(3, {0b001: pi, 0b011: pz, 0b101: px, 0b111: py}, 1 - pi - px - py - pz)


def correlated_error_probs(probabilities) -> np.ndarray:
    """``CORRELATED_ERROR(p1) ELSE_CORRELATED_ERROR(p2) ...``: one bit per branch, at most one fires - branch
    ``i`` with its own probability times the probability that no earlier branch did."""
    fires, none_yet = {}, 1.0
    for branch, p in enumerate(probabilities):
        fires[1 << branch] = none_yet * p
        none_yet *= 1 - p
    return _outcome_table(len(probabilities), fires, none_yet)


# ---- the two arithmetic primitives ---------------------------------------------------------------------------


def xor_convolve(pa: np.ndarray, pb: np.ndarray) -> np.ndarray:
    """Distribution of ``A xor B`` for independent ``A ~ pa``, ``B ~ pb`` over the same outcome space:
    ``out[o] = sum_a pa[a] * pb[a ^ o]``, the terms of every entry added in increasing ``a`` (the accumulation
    order the reference's tables have) - one permuted copy of ``pb`` per ``a``."""
    pa, pb = np.asarray(pa, dtype=np.float64), np.asarray(pb, dtype=np.float64)
    if pa.shape != pb.shape:
        raise ValueError(f"cannot convolve distributions over {len(pa)} and {len(pb)} outcomes")
    partner = np.arange(len(pa))
    out = np.zeros_like(pa)
    for a, weight in enumerate(pa):
        out += weight * pb[partner ^ a]
    return out


def push_forward(probs: np.ndarray, images: list, n_new: int, scan: list | None = None) -> np.ndarray:
    """Push an outcome distribution through a GF(2)-linear map of the outcome bits.

    ``images[i]`` is the image of old bit ``i`` as a bit mask over the ``n_new`` new bits (0: the bit is
    dropped; several old bits may share an image - their contributions XOR).  New outcome ``y`` collects
    ``probs[x]`` over all ``x`` with ``L(x) = y``, visiting the old outcomes in the order of the counter
    ``s = 0, 1, ...`` whose bit ``j`` is old bit ``scan[j]`` (default: ``scan[j] = j``, i.e. increasing ``x``).
    """
    k = len(images)
    scan = list(range(k)) if scan is None else list(scan)
    s = np.arange(1 << k)
    old = np.zeros(1 << k, dtype=np.int64)
    new = np.zeros(1 << k, dtype=np.int64)
    for j, i in enumerate(scan):
        bit = (s >> j) & 1
        old |= bit << i
        new ^= bit * int(images[i])
    out = np.zeros(1 << n_new, dtype=np.float64)
    np.add.at(out, new, np.asarray(probs, dtype=np.float64)[old])  # unbuffered: sequential, in visiting order
    return out


# ---- reduction of a noise model to independent channels on distinct signature sets ------------------------------


def marginalise(channel: Channel, null_col_id: int | None) -> Channel | None:
    """Sum out the bits that drive no ``f`` (signature = the all-zero column); ``None`` if no bit is left.

    Bit ``i`` is axis ``i`` of the Fortran-ordered ``(2,) * k`` view of ``probs``; the summation is numpy's
    multi-axis ``sum`` because its addition order is part of the golden tables."""
    live = [i for i, cid in enumerate(channel.unique_col_ids) if cid != null_col_id]
    if null_col_id is None or len(live) == channel.num_bits:
        return channel
    if not live:
        return None
    k = channel.num_bits
    dead = tuple(i for i in range(k) if i not in live)
    cube = np.asarray(channel.probs).reshape((2,) * k, order="F")
    return Channel(cube.sum(axis=dead).reshape(1 << len(live), order="F"), tuple(channel.unique_col_ids[i] for i in live))


def canonicalise(channel: Channel) -> Channel:
    """Distinct, ascending signature ids: bits are visited in id order (stable) and bits that carry the same
    signature collapse into one (only their parity reaches ``f``) - a single push-forward."""
    ids = channel.unique_col_ids
    support = tuple(sorted(set(ids)))
    slot = {cid: pos for pos, cid in enumerate(support)}
    by_id = sorted(range(len(ids)), key=lambda i: ids[i])  # stable
    probs = push_forward(channel.probs, [1 << slot[cid] for cid in ids], len(support), scan=by_id)
    return Channel(probs, support)


def combine_supports(channels: list, max_bits: int = 4) -> list:
    """Independent channels on the same support XOR-convolve; a channel whose support is strictly inside a
    wider one of at most ``max_bits`` bits is embedded into it and convolved in.

    Order (it fixes the float results): supports in first-appearance order, members convolved left to right;
    then widest supports first (stable); a narrow channel joins the FIRST wider host in that order, hosts
    take their guests in that order too."""
    merged: dict = {}
    for ch in channels:
        have = merged.get(ch.unique_col_ids)
        merged[ch.unique_col_ids] = np.array(ch.probs, dtype=np.float64) if have is None else xor_convolve(have, ch.probs)
    ranked = sorted(merged.items(), key=lambda kv: -len(kv[0]))
    host_of: dict = {}
    for j, (ids_j, _) in enumerate(ranked):
        need = set(ids_j)
        for i in range(j):
            ids_i = ranked[i][0]
            if i not in host_of and len(ids_i) <= max_bits and need < set(ids_i):
                host_of[j] = i
                break
    out = []
    for i, (ids_i, probs_i) in enumerate(ranked):
        if i in host_of:
            continue
        place = {cid: pos for pos, cid in enumerate(ids_i)}
        for j in (g for g, h in host_of.items() if h == i):  # ascending j: dict insertion order
            ids_j, probs_j = ranked[j]
            probs_i = xor_convolve(probs_i, push_forward(probs_j, [1 << place[cid] for cid in ids_j], len(ids_i)))
        out.append(Channel(probs_i, ids_i))
    return out


def simplify_channels(channels: list, max_bits: int = 4, null_col_id: int | None = None) -> list:
    """The independent channels, on distinct signature sets, that reproduce the noise model's ``f`` statistics."""
    kept = (marginalise(ch, null_col_id) for ch in channels)
    return combine_supports([canonicalise(ch) for ch in kept if ch is not None], max_bits)


# ---- signatures ---------------------------------------------------------------------------------------------


def column_signatures(error_transform: np.ndarray):
    """``(signature_matrix uint8[n_sig, num_f], ids int[num_e], null_id)``: distinct columns in lexicographic
    order (row 0 first) and, per error bit, the number of its column."""
    T = np.asarray(error_transform)
    num_f, num_e = (T.shape if T.ndim == 2 else (0, 0))
    if num_e == 0:
        return np.zeros((0, num_f), np.uint8), np.zeros(0, np.int64), None
    bits = np.ascontiguousarray((T != 0).T.astype(np.uint8))  # one row per column
    packed = np.packbits(bits, axis=1)                         # big-endian: row 0 of T is the top bit
    keys = [int.from_bytes(row.tobytes(), "big") for row in packed]
    distinct = sorted(set(keys))
    number = {key: n for n, key in enumerate(distinct)}
    ids = np.fromiter((number[key] for key in keys), dtype=np.int64, count=num_e)
    first = {}
    for col, key in enumerate(keys):
        first.setdefault(key, col)
    signature_matrix = np.stack([bits[first[key]] for key in distinct]).astype(np.uint8)
    null_id = number.get(0)
    return signature_matrix, ids, null_id


def pack_rows(bits: np.ndarray) -> np.ndarray:
    """``uint8[n, num_f]`` (0/1) -> ``uint64[n, ceil(num_f/64)]`` little-endian bit rows (>= 1 word)."""
    n, num_f = bits.shape
    wf = max(1, (num_f + 63) // 64)
    padded = np.zeros((n, wf * 64), dtype=np.uint8)
    padded[:, :num_f] = bits
    return np.packbits(padded, axis=1, bitorder="little").view(np.uint64).reshape(n, wf)


def unpack_rows(rows: np.ndarray, num_f: int) -> np.ndarray:
    """Inverse of :func:`pack_rows`: a fresh, contiguous ``uint8[n, num_f]``."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    bits = np.unpackbits(rows.view(np.uint8).reshape(rows.shape[0], -1), axis=1, bitorder="little")
    return np.ascontiguousarray(bits[:, :num_f])


# ---- the sampler -----------------------------------------------------------------------------------------------


class ChannelSampler:
    """Samples every error channel and maps the error bits to the reduced basis ``f``.

    Attributes mirrored from the reference: ``channels`` (simplified), ``signature_matrix``."""

    def __init__(self, channel_probs: list, error_transform: np.ndarray, seed: int | None = None, engine: str = "auto"):
        """``engine``: ``"numpy"`` draws through ``numpy.random.Generator`` as the reference does; ``"native"`` runs
        the same stream (PCG64, ziggurat exponential, geometric, uniform - bit for bit) inside ``libtsim_hip.so``
        (``tsim_pcg_sample_channels``), an order of magnitude faster; ``"auto"`` = native when the library is
        there and reproduces this numpy's stream (checked once per process), else numpy."""
        if engine not in ("auto", "numpy", "native"):
            raise ValueError("engine must be 'auto', 'numpy' or 'native'")
        self.signature_matrix, ids, null_id = column_signatures(error_transform)
        raw, at = [], 0
        for probs in channel_probs:
            k = int(np.log2(len(probs)))
            raw.append(Channel(probs, tuple(int(v) for v in ids[at:at + k])))
            at += k
        self.channels = simplify_channels(raw, null_col_id=null_id)
        if seed is None:
            seed = np.random.default_rng().integers(0, 2**30)
        self._rng = np.random.default_rng(seed)
        self._sig_words = pack_rows(self.signature_matrix) if len(self.signature_matrix) else np.zeros((0, self.num_words), np.uint64)
        self._tables = self._firing_tables()
        # (p_fire, conditional CDF, uint8 XOR patterns) per firing channel: what the device-side sampler uploads
        self._sparse_data = [(p, cdf, unpack_rows(pats, self.num_f)) for p, cdf, pats in self._tables]
        self._native = None
        if engine != "numpy":
            ok, why = native_stream_available()
            if ok:
                self._native = _NativeTables(self._tables, self.num_words)
            elif engine == "native":
                raise RuntimeError(f"native channel sampler unavailable: {why}")

    @property
    def num_f(self) -> int:
        return int(self.signature_matrix.shape[1])

    @property
    def num_words(self) -> int:
        return max(1, (self.num_f + 63) // 64)

    def _firing_tables(self) -> list:
        """Per channel that can fire: ``(p_fire, cdf over the non-identity outcomes, packed XOR patterns)``.

        The float sequence (``1 - p0``, ``cumsum(p[1:] / p_fire)``, renormalisation by the last entry) decides
        which outcome a uniform draw selects and is therefore the reference's (channels.py:600-610)."""
        tables = []
        for ch in self.channels:
            probs = np.asarray(ch.probs, dtype=np.float64)
            p_fire = 1.0 - float(probs[0])
            if len(probs) <= 1 or p_fire <= 1e-15:
                continue
            cdf = np.cumsum(probs[1:] / p_fire, dtype=np.float64)
            cdf /= cdf[-1]
            patterns = np.zeros((len(probs) - 1, self.num_words), dtype=np.uint64)
            outcomes = np.arange(1, len(probs))
            for bit, cid in enumerate(ch.unique_col_ids):
                hit = ((outcomes >> bit) & 1).astype(bool)
                patterns[hit] ^= self._sig_words[cid]
            tables.append((p_fire, cdf, patterns))
        return tables

    def _draw(self, p_fire: float, cdf: np.ndarray, num_samples: int):
        """Geometric-skip draw of one channel: ``(rows that fire, outcome index per fired row)``.

        The generator calls and their sizes are the stream contract (channels.py:641-655): ``n_draws``
        geometric gaps (7 sigma above the expected number of fires, + 100), then one uniform per fired row
        that lies inside the batch."""
        expected = num_samples * p_fire
        n_draws = int(expected + 7.0 * np.sqrt(expected * (1.0 - p_fire))) + 100
        rows = np.cumsum(self._rng.geometric(p_fire, size=n_draws)) - 1
        rows = rows[rows < num_samples]
        if len(rows) == 0:
            return rows, None
        return rows, np.searchsorted(cdf, self._rng.uniform(size=len(rows)))

    def sample_packed(self, num_samples: int = 1, out: np.ndarray | None = None) -> np.ndarray:
        """``uint64[num_samples, ceil(num_f/64)]``: the rows the sampling kernels read (bit ``i`` of a row =
        ``f_i``).  Fired rows of one channel are distinct, so the scatter is a plain indexed XOR per word.
        ``out``: optional C-contiguous destination of that shape (a reused staging buffer saves the page faults
        of a fresh 8 MB array per batch)."""
        if out is not None and (out.shape != (num_samples, self.num_words) or out.dtype != np.uint64 or not out.flags.c_contiguous):
            raise ValueError("out must be a C-contiguous uint64 array of shape (num_samples, num_words)")
        if self._native is not None and 0 < num_samples < (1 << 32):
            return self._native.sample(self._rng, num_samples, out)
        if out is None:
            out = np.zeros((num_samples, self.num_words), dtype=np.uint64)
        else:
            out[...] = 0
        for p_fire, cdf, patterns in self._tables:
            rows, outcome = self._draw(p_fire, cdf, num_samples)
            if outcome is None:
                continue
            for w in range(self.num_words):
                column = patterns[:, w]
                if column.any():
                    out[rows, w] ^= column[outcome]
        return out

    def sample(self, num_samples: int = 1) -> np.ndarray:
        """``uint8[num_samples, num_f]``, the reference's layout - same generator stream as :meth:`sample_packed`
        (the two can be mixed call by call)."""
        return unpack_rows(self.sample_packed(num_samples), self.num_f)


# ---- the same stream, natively ------------------------------------------------------------------------------------


class _PcgState(C.Structure):
    _fields_ = [("state_lo", C.c_uint64), ("state_hi", C.c_uint64), ("inc_lo", C.c_uint64), ("inc_hi", C.c_uint64)]


_M64 = (1 << 64) - 1


def _export_state(gen: np.random.Generator) -> tuple:
    st = gen.bit_generator.state
    if st.get("bit_generator") != "PCG64":
        raise TypeError("the native sampler reproduces PCG64 streams only")
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    return st, _PcgState(s & _M64, s >> 64, inc & _M64, inc >> 64)


def _import_state(gen: np.random.Generator, st: dict, raw: _PcgState) -> None:
    st["state"]["state"] = (int(raw.state_hi) << 64) | int(raw.state_lo)
    gen.bit_generator.state = st  # the 32-bit half-word buffer (has_uint32 / uinteger) is untouched: never used here


class _NativeTables:
    """The firing tables in the flat layout ``tsim_pcg_sample_channels`` takes."""

    def __init__(self, tables: list, words: int):
        from . import _lib

        self._fn = _lib.load().tsim_pcg_sample_channels
        self._check = _lib.check
        self.words = int(words)
        self.n = len(tables)
        self.p_fire = np.ascontiguousarray([t[0] for t in tables], dtype=np.float64)
        self.n_out = np.ascontiguousarray([len(t[1]) for t in tables], dtype=np.int32)
        self.cdf = np.ascontiguousarray(np.concatenate([t[1] for t in tables]) if tables else np.zeros(0), dtype=np.float64)
        self.patterns = np.ascontiguousarray(np.concatenate([t[2] for t in tables], axis=0) if tables
                                             else np.zeros((0, self.words)), dtype=np.uint64)

    def sample(self, gen: np.random.Generator, num_samples: int, out: np.ndarray | None = None) -> np.ndarray:
        st, raw = _export_state(gen)
        if out is None:
            out = np.empty((num_samples, self.words), dtype=np.uint64)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._check(self._fn(C.byref(raw), self.n, ptr(self.p_fire), ptr(self.n_out), ptr(self.cdf), ptr(self.patterns),
                             self.words, int(num_samples), ptr(out), 0), "tsim_pcg_sample_channels")
        _import_state(gen, st, raw)
        return out


_native_verdict: tuple | None = None


def native_stream_available() -> tuple:
    """``(ok, reason)``: the library loads and its PCG64 / ziggurat / geometric / uniform draws equal this
    numpy's, draw for draw, on a seeded stream long enough to visit the ziggurat's rare branches.  The tables
    inside the library were measured from one numpy version (csrc/tsim_zig_tables.h); another numpy with other
    constants is detected here and the numpy engine is used instead - the drop-in contract is the numpy stream."""
    global _native_verdict
    if _native_verdict is not None:
        return _native_verdict
    try:
        from . import _lib

        # (a probe: constructing a host-side sampler must never start a multi-minute build of the GPU library)
        lib = _lib.load(build=False)
        gen = np.random.default_rng(20260928)
        twin = np.random.default_rng(20260928)
        for kind, p, n, dtype, draw in (
            (2, 0.0, 200_000, np.float64, lambda g, n: g.standard_exponential(n)),
            (3, 0.02, 50_000, np.int64, lambda g, n: g.geometric(0.02, n)),
            (3, 0.5, 20_000, np.int64, lambda g, n: g.geometric(0.5, n)),
            (1, 0.0, 20_000, np.float64, lambda g, n: g.uniform(size=n)),
            (3, 1e-7, 5_000, np.int64, lambda g, n: g.geometric(1e-7, n)),
        ):
            want = draw(gen, n)
            st, raw = _export_state(twin)
            got = np.empty(n, dtype=dtype)
            _lib.check(lib.tsim_pcg_draw(C.byref(raw), kind, float(p), n, got.ctypes.data_as(C.c_void_p)), "tsim_pcg_draw")
            _import_state(twin, st, raw)
            if not np.array_equal(want, got) or twin.bit_generator.state != gen.bit_generator.state:
                _native_verdict = (False, f"native draws of kind {kind} (p = {p}) differ from numpy {np.__version__}")
                warnings.warn("tsim_amd: " + _native_verdict[1] + "; falling back to the numpy channel sampler", stacklevel=2)
                return _native_verdict
        _native_verdict = (True, "")
    except Exception as exc:  # library missing / not loadable: the numpy engine still gives the right stream
        _native_verdict = (False, repr(exc))
        warnings.warn(f"tsim_amd: native channel sampler unavailable ({exc!r}); using the numpy engine - the same stream, "
                      "about 10x slower (build the library with `python -m tsim_amd.build`)", stacklevel=2)
    return _native_verdict
