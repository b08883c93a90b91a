"""Host-side noise sampling: error channels -> error-mechanism bit-vectors ``f``.

Mirror of the reference's ``ChannelSampler`` (src/tsim/noise/channels.py:503-658) with the
same constructor arguments, attributes (``channels``, ``signature_matrix``) and ``sample``
semantics, so that for a fixed seed it consumes numpy's PCG64 stream identically and returns
the same ``uint8[num_samples, num_f]`` array:

* ``f_i = XOR_j error_transform[i, j] * e_j``;
* at construction the channels are simplified - bits that touch no ``f`` are marginalised,
  bits with equal column signatures are XOR-folded, channels over the same signature set are
  XOR-convolved, and channels whose signature set is a strict subset of another's (of at most
  ``max_bits`` bits) are absorbed into it (src/tsim/noise/channels.py:230-500);
* sampling is geometric-skip per channel: positions of firing shots from cumulative
  ``Generator.geometric`` draws, outcome by inverse-CDF on ``Generator.uniform``, then the
  outcome's precomputed XOR pattern is applied (src/tsim/noise/channels.py:578-658).

This is host logic of the path, written against numpy only.  It is validated against golden
vectors produced by the reference module itself (tests/golden/gen_channels_golden.py).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Channel:
    """A distribution over the 2^k outcomes of k error bits.

    ``probs[o]`` is the probability of outcome ``o`` (bit i of ``o`` = error bit i);
    ``unique_col_ids[i]`` names the error-transform column signature that bit i drives.
    """

    probs: np.ndarray
    unique_col_ids: tuple

    def __post_init__(self) -> None:
        tol = 1e-6
        if np.any(self.probs < -tol) or np.any(self.probs > 1.0 + tol):
            raise ValueError(f"Probabilities must lie in [0, 1], but got: {self.probs}")
        if not np.isclose(np.sum(self.probs), 1.0):
            raise ValueError(
                f"Probabilities must sum to 1, but got: {self.probs} (sum {np.sum(self.probs)})"
            )

    @property
    def num_bits(self) -> int:
        return int(np.log2(len(self.probs)))


# ---- probability constructors (src/tsim/noise/channels.py:47-205) ----------


def error_probs(p: float) -> np.ndarray:
    """One-bit channel ``[1-p, p]``."""
    return np.array([1 - p, p], dtype=np.float64)


def pauli_channel_1_probs(px: float, py: float, pz: float) -> np.ndarray:
    """Single-qubit Pauli channel; bit 0 = Z component, bit 1 = X component."""
    return np.array([1 - px - py - pz, pz, px, py], dtype=np.float64)


def heralded_pauli_channel_1_probs(pi: float, px: float, py: float, pz: float) -> np.ndarray:
    """Heralded Pauli channel; bit 0 = herald, bit 1 = Z, bit 2 = X."""
    probs = np.zeros(8, dtype=np.float64)
    probs[0] = 1 - pi - px - py - pz
    probs[0b001], probs[0b011], probs[0b101], probs[0b111] = pi, pz, px, py
    return probs


def correlated_error_probs(probabilities) -> np.ndarray:
    """``CORRELATED_ERROR(p1) ELSE_CORRELATED_ERROR(p2) ...``: at most one bit fires."""
    k = len(probabilities)
    probs = np.zeros(2**k, dtype=np.float64)
    survive = 1.0
    for i, p in enumerate(probabilities):
        probs[1 << i] = survive * p
        survive *= 1 - p
    probs[0] = survive
    return probs


# ---- simplification passes ----------------------------------------------------


def xor_convolve(pa: np.ndarray, pb: np.ndarray) -> np.ndarray:
    """``P(A xor B = o) = sum_{a ^ b = o} P(A=a) P(B=b)`` (accumulated in (a, b) order)."""
    n = len(pa)
    if len(pb) != n:
        raise ValueError("Both channels must have same number of outcomes")
    out = np.zeros(n, dtype=np.float64)
    for a in range(n):
        for b in range(n):
            out[a ^ b] += pa[a] * pb[b]
    return out


def _remap_outcomes(probs: np.ndarray, bit_targets: list, n_new: int) -> np.ndarray:
    """Push every outcome through a GF(2)-linear bit map and accumulate in outcome order.

    ``bit_targets[i]`` is the new bit position driven by old bit i (``None`` = dropped).
    Several old bits may drive the same new bit (their contributions XOR).
    """
    out = np.zeros(2**n_new, dtype=np.float64)
    for old in range(len(probs)):
        new = 0
        for i, tgt in enumerate(bit_targets):
            if tgt is not None and (old >> i) & 1:
                new ^= 1 << tgt
        out[new] += probs[old]
    return out


def reduce_null_bits(channels: list, null_col_id: int | None = None) -> list:
    """Marginalise bits that map to the all-zero column; drop channels left with no bits."""
    if null_col_id is None:
        return channels
    result = []
    for ch in channels:
        keep = [i for i, cid in enumerate(ch.unique_col_ids) if cid != null_col_id]
        if not keep:
            continue
        n = ch.num_bits
        if len(keep) == n:
            result.append(Channel(probs=ch.probs, unique_col_ids=tuple(ch.unique_col_ids)))
            continue
        drop = tuple(i for i in range(n) if i not in keep)
        # little-endian bit i == axis i of the Fortran-ordered (2,)*n tensor
        tensor = ch.probs.reshape((2,) * n, order="F")
        new_probs = tensor.sum(axis=drop).reshape(2 ** len(keep), order="F")
        result.append(Channel(probs=new_probs, unique_col_ids=tuple(ch.unique_col_ids[i] for i in keep)))
    return result


def normalize_channels(channels: list) -> list:
    """Sort each channel's column ids (stable) and permute its outcome bits accordingly."""
    result = []
    for ch in channels:
        ids = np.array(ch.unique_col_ids)
        perm = np.argsort(ids, stable=True)  # new bit j <- old bit perm[j]
        n = ch.num_bits
        new_probs = np.empty_like(ch.probs)
        for old in range(len(ch.probs)):
            new = 0
            for j in range(n):
                new |= ((old >> int(perm[j])) & 1) << j
            new_probs[new] = ch.probs[old]
        result.append(Channel(probs=new_probs, unique_col_ids=tuple(ids[perm])))
    return result


def fold_duplicate_channel_bits(channels: list) -> list:
    """Bits of one channel with the same column id only matter through their parity."""
    result = []
    for ch in channels:
        uniq = tuple(dict.fromkeys(ch.unique_col_ids))
        if len(uniq) == len(ch.unique_col_ids):
            result.append(ch)
            continue
        pos = {c: i for i, c in enumerate(uniq)}
        targets = [pos[c] for c in ch.unique_col_ids]
        result.append(Channel(probs=_remap_outcomes(ch.probs, targets, len(uniq)), unique_col_ids=uniq))
    return result


def expand_channel(channel: Channel, target_col_ids: tuple) -> Channel:
    """Embed a channel into a strict superset of (sorted, duplicate-free) column ids."""
    src = channel.unique_col_ids
    if src != tuple(sorted(src)):
        raise ValueError("Source must be sorted")
    if target_col_ids != tuple(sorted(target_col_ids)):
        raise ValueError("Target must be sorted")
    if len(set(target_col_ids)) != len(target_col_ids):
        raise ValueError("Target must not contain duplicates")
    if not set(src) < set(target_col_ids):
        raise ValueError("Source must be strict subset")
    targets = [target_col_ids.index(s) for s in src]
    return Channel(
        probs=_remap_outcomes(channel.probs, targets, len(target_col_ids)), unique_col_ids=target_col_ids
    )


def merge_identical_channels(channels: list) -> list:
    """XOR-convolve all channels over the same signature tuple (first-seen group order)."""
    groups: dict = {}
    for ch in channels:
        groups.setdefault(ch.unique_col_ids, []).append(ch)
    result = []
    for ids, grp in groups.items():
        if len(grp) == 1:
            result.append(grp[0])
            continue
        acc = grp[0].probs.copy()
        for ch in grp[1:]:
            acc = xor_convolve(acc, ch.probs)
        result.append(Channel(probs=acc, unique_col_ids=ids))
    return result


def absorb_subset_channels(channels: list, max_bits: int = 4) -> list:
    """Fold every channel whose signature set is a strict subset of a larger one (<= max_bits) into it."""
    order = sorted(channels, key=lambda c: -len(c.unique_col_ids))  # stable, widest first
    gone: set = set()
    result = []
    for i, big in enumerate(order):
        if i in gone:
            continue
        big_set = set(big.unique_col_ids)
        probs = big.probs.copy()
        for j in range(i + 1, len(order)):
            if j in gone:
                continue
            small = order[j]
            if set(small.unique_col_ids) < big_set and len(big_set) <= max_bits:
                probs = xor_convolve(probs, expand_channel(small, big.unique_col_ids).probs)
                gone.add(j)
        result.append(Channel(probs=probs, unique_col_ids=big.unique_col_ids))
    return result


def simplify_channels(channels: list, max_bits: int = 4, null_col_id: int | None = None) -> list:
    """null-bit removal -> sort -> fold duplicates -> merge identical -> absorb subsets."""
    channels = reduce_null_bits(channels, null_col_id)
    channels = normalize_channels(channels)
    channels = fold_duplicate_channel_bits(channels)
    channels = merge_identical_channels(channels)
    channels = absorb_subset_channels(channels, max_bits)
    return channels


# ---- the sampler ----------------------------------------------------------------


class ChannelSampler:
    """Samples all error channels and maps the error bits to the reduced ``f`` basis."""

    def __init__(self, channel_probs: list, error_transform: np.ndarray, seed: int | None = None):
        error_transform = np.asarray(error_transform)
        unique_cols, inverse = np.unique(error_transform, axis=1, return_inverse=True)
        inverse = np.asarray(inverse).reshape(-1)
        signature_matrix = unique_cols.T  # one row per distinct column signature
        zero_cols = np.flatnonzero(np.all(unique_cols == 0, axis=0))
        null_col_id = int(zero_cols[0]) if len(zero_cols) else None

        channels = []
        offset = 0
        for probs in channel_probs:
            k = int(np.log2(len(probs)))
            ids = tuple(int(inverse[offset + i]) for i in range(k))
            channels.append(Channel(probs=probs, unique_col_ids=ids))
            offset += k

        self.channels = simplify_channels(channels, null_col_id=null_col_id)
        self.signature_matrix = signature_matrix.astype(np.uint8)
        self._rng = np.random.default_rng(
            seed if seed is not None else np.random.default_rng().integers(0, 2**30)
        )
        self._sparse_data = self._precompute_sparse(self.channels, self.signature_matrix)
        self._packed_patterns = None  # built on first use by sample_packed

    @property
    def num_f(self) -> int:
        return int(self.signature_matrix.shape[1])

    @staticmethod
    def _precompute_sparse(channels: list, signature_matrix: np.ndarray) -> list:
        """Per channel: ``(p_fire, conditional CDF over non-identity outcomes, XOR patterns)``."""
        data = []
        for ch in channels:
            probs = ch.probs.astype(np.float64)
            p_fire = 1.0 - float(probs[0])
            n_outcomes = len(probs)
            if p_fire <= 1e-15 or n_outcomes <= 1:
                continue
            cond_cdf = np.cumsum(probs[1:] / p_fire, dtype=np.float64)
            cond_cdf /= cond_cdf[-1]
            ids = np.asarray(ch.unique_col_ids)
            k = len(ids)
            outcomes = np.arange(1, n_outcomes)
            bits = ((outcomes[:, None] >> np.arange(k)) & 1).astype(np.uint8)
            xor_patterns = (bits @ signature_matrix[ids] % 2).astype(np.uint8)
            data.append((p_fire, cond_cdf, xor_patterns))
        return data

    def sample(self, num_samples: int = 1) -> np.ndarray:
        """``uint8[num_samples, num_f]`` (geometric-skip sampling, channels.py:624-658)."""
        result = np.zeros((num_samples, self.signature_matrix.shape[1]), dtype=np.uint8)
        for p_fire, cond_cdf, xor_pats in self._sparse_data:
            expected = num_samples * p_fire
            sigma = np.sqrt(expected * (1.0 - p_fire))
            n_draws = int(expected + 7.0 * sigma) + 100  # 7 sigma: undersampling ~1e-12
            positions = np.cumsum(self._rng.geometric(p_fire, size=n_draws)) - 1
            positions = positions[positions < num_samples]
            if len(positions) == 0:
                continue
            outcome = np.searchsorted(cond_cdf, self._rng.uniform(size=len(positions)))
            result[positions] ^= xor_pats[outcome]
        return result

    def sample_packed(self, num_samples: int = 1) -> np.ndarray:
        """The same samples as :meth:`sample` (same generator calls in the same order, so the two are
        interchangeable mid-stream) as packed rows ``uint64[num_samples, ceil(num_f/64)]`` - what the
        sampling kernel reads.  The XOR scatter touches 8 bytes per fired row and word instead of
        ``num_f`` bytes, which is where :meth:`sample` spends its time (0.56 s of 0.69 s per 10^6
        shots x 64 one-bit channels at p = 0.02)."""
        num_f = self.signature_matrix.shape[1]
        wf = max(1, (num_f + 63) // 64)
        rows = np.zeros((num_samples, wf), dtype=np.uint64)
        if self._packed_patterns is None:
            packed = []
            for _, _, xor_pats in self._sparse_data:
                pad = np.zeros((xor_pats.shape[0], wf * 64), dtype=np.uint8)
                pad[:, :num_f] = xor_pats
                packed.append(np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(-1, wf))
            self._packed_patterns = packed
        for (p_fire, cond_cdf, _), pats in zip(self._sparse_data, self._packed_patterns):
            expected = num_samples * p_fire
            sigma = np.sqrt(expected * (1.0 - p_fire))
            n_draws = int(expected + 7.0 * sigma) + 100
            positions = np.cumsum(self._rng.geometric(p_fire, size=n_draws)) - 1
            positions = positions[positions < num_samples]
            if len(positions) == 0:
                continue
            outcome = np.searchsorted(cond_cdf, self._rng.uniform(size=len(positions)))
            for w in range(wf):  # positions are strictly increasing, hence unique: plain fancy XOR
                col = pats[:, w]
                if col.any():
                    rows[positions, w] ^= col[outcome]
        return rows
