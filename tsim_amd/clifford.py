"""Clifford-only circuit front-end: Stim-style text -> compiled detector-sampling program.

SURVEY.md section 8(f) row 4.  For a circuit made of Clifford gates, resets, measurements and
Pauli noise every detector/observable is ``flip XOR (XOR of some error bits e_i)``; the reference
finds that by ZX reduction (``prepare_graph`` -> ``zx.full_reduce``, src/tsim/core/graph.py:517-557),
here it comes from **Pauli-frame propagation** (which errors flip which measurement) plus a small
**stabilizer-tableau run** of the noiseless circuit (the constant ``flip``).  From there on the
reference's own conventions are followed so that the resulting program, channel list and
``error_transform`` are the ones the reference would hand to its sampler:

* error variables and channel tables in circuit order, Z component before X component
  (src/tsim/core/instructions.py:620-724, measurement errors :818-840);
* outputs = detectors in order, then observables by index (core/graph.py:287-303);
* ``e -> f`` change of basis by the greedy GF(2) elimination of ``find_basis``
  (src/tsim/utils/linalg.py:8-79) over the outputs' error sets in output order
  (``transform_error_basis``, core/graph.py:312-381);
* an output that is a single ``f`` bit is a *direct* entry ``f[idx] ^ flip``
  (``classify_direct``, core/graph.py:68-124; ``compile_program``, compile/pipeline.py:55-102),
  sampled by ``_sample_direct`` (sampler.py:547-555).

Outputs that are XORs of several basis bits (linearly dependent detectors) become one-output
components whose marginal is a 0/1 delta (a NodePhases factor over the f bits and the outcome bit);
outputs that no error touches read an extra, always-zero ``f`` column.  Both give the bits the
reference gives; the *structure* of those rare components is this front-end's own.

Measurement sampling (``compile_sampler``): a record is ``const XOR <S, r> XOR <E, e>`` with ``r`` the
random outcomes of the noiseless run (independent uniform bits).  Records that share random bits
form a component; inside it an output is either free (marginal 1/2) or fixed by earlier outputs, so
level k is ONE graph: a delta (NodePhases factor) per fixed output so far times 2^-k.  The marginals
are exactly those of the reference's compiled graphs (0, 1/2 or 1 of the previous level), so the
Threefry draws reproduce its seeded samples (``H 0; M 0`` -> 48, 53, 52, 50); with several components
their processing order (size, then first output) is this front-end's reading of pipeline.py:65.

Gate coverage: every Clifford entry of the reference's gate table (core/instructions.py GATE_TABLE) -
the 24 single-qubit Cliffords (Paulis, H_*, S/SQRT_*, the C_* axis cycles), all A-controlled-B gates,
SQRT_XX/YY/ZZ(+_DAG), SWAP/ISWAP/CXSWAP/SWAPCX/CZSWAP, SPP/SPP_DAG Pauli-product phases - defined by what
they do (conjugation tables / controlled-Pauli and sqrt-of-Pauli-product constructions) and checked in
tests/test_clifford.py against unitaries written from those definitions.

Not covered (``NotImplementedError``): non-Clifford gates (T, ``S[T]``, ``I[R_Z(...)]``, ``U3`` and
``SPP[T]`` tags - they need the stabilizer-rank compiler, out of scope) and ``sweep[]`` targets.
Classically controlled Paulis (``CX rec[-1] 2``, ``CZ``, ``CY``, ``XCZ``/``YCZ`` with the record last) are
applied to frame and tableau alike.
"""

from __future__ import annotations

import re
from dataclasses import dataclass, field

import numpy as np

from .channels import correlated_error_probs, error_probs, heralded_pauli_channel_1_probs, pauli_channel_1_probs
from .program import CompiledComponent, CompiledProgram, make_program, scalar_graphs_from_terms

__all__ = ["CliffordCircuit", "find_basis", "pauli_channel_2_probs"]


# ---------------------------------------------------------------------------
# channel table of the two-qubit Pauli channel (src/tsim/noise/channels.py:97-167):
# index = z_i + 2 x_i + 4 z_j + 8 x_j
# ---------------------------------------------------------------------------
_P2_ARGS = ("IX", "IY", "IZ", "XI", "XX", "XY", "XZ", "YI", "YX", "YY", "YZ", "ZI", "ZX", "ZY", "ZZ")
_XZ = {"I": (0, 0), "X": (1, 0), "Y": (1, 1), "Z": (0, 1)}


def pauli_channel_2_probs(*p15: float) -> np.ndarray:
    """``PAULI_CHANNEL_2`` arguments in Stim's order (IX, IY, ..., ZZ) -> 16-entry table."""
    if len(p15) != 15:
        raise ValueError("PAULI_CHANNEL_2 takes 15 probabilities")
    probs = np.zeros(16, dtype=np.float64)
    probs[0] = 1.0 - float(np.sum(np.asarray(p15, dtype=np.float64)))
    for name, p in zip(_P2_ARGS, p15):
        (xi, zi), (xj, zj) = _XZ[name[0]], _XZ[name[1]]
        probs[zi + 2 * xi + 4 * zj + 8 * xj] = p
    return probs


# ---------------------------------------------------------------------------
# GF(2) basis (restates src/tsim/utils/linalg.py:8-79 on Python-int bit rows)
# ---------------------------------------------------------------------------
def find_basis(rows: list[int]) -> tuple[list[int], list[int]]:
    """Greedy basis of bit-vector rows: ``(basis_row_indices, combos)``.

    A row is a basis row iff it is independent of the rows before it; ``combos[i]`` is the bitmask
    (over basis positions) of the basis rows whose XOR gives row ``i``.
    """
    reduced: list[tuple[int, int, int]] = []  # (reduced vector, pivot bit, expansion over basis positions)
    basis_idx: list[int] = []
    combos: list[int] = []
    for i, v in enumerate(rows):
        expansion = 0
        for rv, pivot, exp in reduced:
            if v & pivot:
                v ^= rv
                expansion ^= exp
        if v:
            pos = len(basis_idx)
            basis_idx.append(i)
            expansion ^= 1 << pos
            reduced.append((v, v & -v, expansion))  # pivot: lowest set bit == np.argmax of the 0/1 row
            combos.append(1 << pos)
        else:
            combos.append(expansion)
    return basis_idx, combos


# ---------------------------------------------------------------------------
# text -> flat instruction list
# ---------------------------------------------------------------------------
@dataclass
class _Instr:
    name: str
    args: tuple
    targets: tuple  # of str tokens
    tag: str = ""


_LINE = re.compile(r"^([A-Za-z_][A-Za-z0-9_]*)(?:\[([^\]]*)\])?(?:\(([^)]*)\))?\s*(.*)$")
# tags the reference gives a non-Clifford meaning to (core/parse.py:31-35, 263-276): S[T], I[R_Z(...)], ...
_NON_CLIFFORD_TAG = re.compile(r"^\s*(T|R_X|R_Y|R_Z|R_PAULI|U3)\b")


def _parse(text: str) -> list[_Instr]:
    lines = [ln.split("#", 1)[0].strip() for ln in text.replace(";", "\n").splitlines()]

    def block(pos: int, depth: int) -> tuple[list[_Instr], int]:
        out: list[_Instr] = []
        while pos < len(lines):
            ln = lines[pos]
            pos += 1
            if not ln:
                continue
            if ln == "}":
                if depth == 0:
                    raise ValueError("unmatched '}'")
                return out, pos
            m = _LINE.match(ln)
            if not m:
                raise ValueError(f"cannot parse line {ln!r}")
            name, tag, args, rest = m.group(1).upper(), m.group(2) or "", m.group(3), m.group(4).strip()
            if name in ("T", "T_DAG") or _NON_CLIFFORD_TAG.match(tag):
                raise NotImplementedError(f"{ln!r} is a non-Clifford gate: the Clifford front-end cannot compile it")
            if name == "REPEAT":
                if not rest.endswith("{"):
                    raise ValueError("REPEAT needs a '{' on the same line")
                count = int(rest[:-1].strip())
                body, pos = block(pos, depth + 1)
                out.extend(body * count)
                continue
            a = tuple(float(x) for x in args.split(",")) if args and args.strip() else ()
            out.append(_Instr(name, a, tuple(rest.split()), tag))
        if depth:
            raise ValueError("missing '}'")
        return out, pos

    return block(0, 0)[0]


# ---------------------------------------------------------------------------
# stabilizer tableau (Aaronson-Gottesman), noiseless reference run
# ---------------------------------------------------------------------------
class _Tableau:
    def __init__(self, n: int):
        self.n = n
        self.x = np.zeros((2 * n + 1, n), dtype=np.uint8)
        self.z = np.zeros((2 * n + 1, n), dtype=np.uint8)
        self.r = np.zeros(2 * n + 1, dtype=np.uint8)
        # random measurement outcomes are symbols: sym[row] = bitmask of the symbols in the row's sign
        self.sym = [0] * (2 * n + 1)
        self.n_random = 0
        for i in range(n):
            self.x[i, i] = 1          # destabilizers X_i
            self.z[n + i, i] = 1      # stabilizers Z_i  (|0...0>)

    def h(self, a):
        self.r ^= self.x[:, a] & self.z[:, a]
        self.x[:, a], self.z[:, a] = self.z[:, a].copy(), self.x[:, a].copy()

    def s(self, a):
        self.r ^= self.x[:, a] & self.z[:, a]
        self.z[:, a] ^= self.x[:, a]

    def cx(self, a, b):
        self.r ^= self.x[:, a] & self.z[:, b] & (self.x[:, b] ^ self.z[:, a] ^ 1)
        self.x[:, b] ^= self.x[:, a]
        self.z[:, a] ^= self.z[:, b]

    def pauli(self, a, px, pz):  # conjugation by X^px Z^pz on qubit a: sign flips only
        if px:
            self.r ^= self.z[:, a]
        if pz:
            self.r ^= self.x[:, a]

    def _rowsum(self, h, i):  # row h <- row h * row i
        x1, z1, x2, z2 = self.x[i].astype(np.int64), self.z[i].astype(np.int64), self.x[h].astype(np.int64), self.z[h].astype(np.int64)
        g = np.where((x1 == 1) & (z1 == 1), z2 - x2, 0)
        g = g + np.where((x1 == 1) & (z1 == 0), z2 * (2 * x2 - 1), 0)
        g = g + np.where((x1 == 0) & (z1 == 1), x2 * (1 - 2 * z2), 0)
        tot = (2 * int(self.r[h]) + 2 * int(self.r[i]) + int(g.sum())) % 4
        self.r[h] = 1 if tot == 2 else 0
        self.sym[h] ^= self.sym[i]
        self.x[h] ^= self.x[i]
        self.z[h] ^= self.z[i]

    def pauli_if(self, a, sym_mask, px=1, pz=0):
        """X^px Z^pz on qubit a controlled by the XOR of the random symbols in sym_mask (reset after a
        random measurement; classically controlled Paulis): rows that anticommute with it pick the
        symbols up in their sign."""
        if sym_mask:
            anti = (self.z[:, a] if px else 0) ^ (self.x[:, a] if pz else 0)
            for row in np.nonzero(anti)[0]:
                self.sym[int(row)] ^= sym_mask

    def measure_z(self, a) -> tuple[int, int]:
        """Z-basis measurement -> (constant part, symbol mask).  A random outcome is a fresh symbol
        (value 0 in the reference run); later outcomes that repeat it carry the same symbol, so a
        detector is deterministic exactly when its symbols cancel."""
        n = self.n
        ps = np.nonzero(self.x[n:2 * n, a])[0]
        if len(ps):
            p = int(ps[0]) + n
            for i in range(2 * n):
                if i != p and self.x[i, a]:
                    self._rowsum(i, p)
            self.x[p - n], self.z[p - n], self.r[p - n] = self.x[p].copy(), self.z[p].copy(), self.r[p]
            self.sym[p - n] = self.sym[p]
            self.x[p] = 0
            self.z[p] = 0
            self.z[p, a] = 1
            self.r[p] = 0
            self.sym[p] = 1 << self.n_random
            self.n_random += 1
            return 0, self.sym[p]
        s = 2 * n
        self.x[s] = 0
        self.z[s] = 0
        self.r[s] = 0
        self.sym[s] = 0
        for i in range(n):
            if self.x[i, a]:
                self._rowsum(s, i + n)
        return int(self.r[s]), self.sym[s]


# ---------------------------------------------------------------------------
# the analysis: frames + tableau driven by one gate decomposition
# ---------------------------------------------------------------------------
# Single-qubit Cliffords are given by what they do (Stim's conventions): the images of X and of Z under
# conjugation.  Each becomes a word over the primitives  H | S | P(x,z) (Pauli X^x Z^z: signs only),
# found once at import time: the six classes modulo Paulis are the words below, the Pauli fixes the signs.
_ACTION_1Q = {
    "I": ("+X", "+Z"), "X": ("+X", "-Z"), "Y": ("-X", "-Z"), "Z": ("-X", "+Z"),
    "H": ("+Z", "+X"), "H_XZ": ("+Z", "+X"), "H_NXZ": ("-Z", "-X"),
    "H_XY": ("+Y", "-Z"), "H_NXY": ("-Y", "-Z"),
    "H_YZ": ("-X", "+Y"), "H_NYZ": ("-X", "-Y"),
    "S": ("+Y", "+Z"), "SQRT_Z": ("+Y", "+Z"), "S_DAG": ("-Y", "+Z"), "SQRT_Z_DAG": ("-Y", "+Z"),
    "SQRT_X": ("+X", "-Y"), "SQRT_X_DAG": ("+X", "+Y"),
    "SQRT_Y": ("-Z", "+X"), "SQRT_Y_DAG": ("+Z", "-X"),
    # period-3 axis cycles, named by the cycle: C_XYZ sends X -> Y -> Z -> X, C_NXYZ sends -X -> Y -> Z -> -X
    "C_XYZ": ("+Y", "+X"), "C_NXYZ": ("-Y", "-X"), "C_XNYZ": ("-Y", "+X"), "C_XYNZ": ("+Y", "-X"),
    "C_ZYX": ("+Z", "+Y"), "C_NZYX": ("-Z", "-Y"), "C_ZNYX": ("+Z", "-Y"), "C_ZYNX": ("-Z", "+Y"),
}


def _solve_one_qubit_words():
    def conj(word, x, z, s):  # the tableau's update rules (Tableau.h / .s / .pauli) on one signed Pauli
        for step in word:
            if step[0] == "H":
                s ^= x & z
                x, z = z, x
            elif step[0] == "S":
                s ^= x & z
                z ^= x
            else:
                s ^= (z & step[1]) ^ (x & step[2])
        return x, z, s

    H, S = ("H",), ("S",)
    classes = [(), (H,), (S,), (H, S), (S, H), (H, S, H)]
    words = {}
    for name, images in _ACTION_1Q.items():
        want = [(_XZ[img[1]][0], _XZ[img[1]][1], int(img[0] == "-")) for img in images]
        for w in classes:
            for px in (0, 1):
                for pz in (0, 1):
                    word = w + ((("P", px, pz),) if (px or pz) else ())
                    if [conj(word, 1, 0, 0), conj(word, 0, 1, 0)] == want:
                        words.setdefault(name, word)
        if name not in words:
            raise AssertionError(f"no Clifford word for {name}")
    return words


_ONE_QUBIT = _solve_one_qubit_words()

# Two-qubit Cliffords by construction: A-controlled-B gates, square roots of two-qubit Pauli products
# ("phase the -1 eigenspace of PP by i"), and SWAP composites.
_CONTROLLED = {"CX": "ZX", "CNOT": "ZX", "ZCX": "ZX", "CY": "ZY", "ZCY": "ZY", "CZ": "ZZ", "ZCZ": "ZZ",
               "XCX": "XX", "XCY": "XY", "XCZ": "XZ", "YCX": "YX", "YCY": "YY", "YCZ": "YZ"}
_SQRT_PP = {"SQRT_XX": ("X", False), "SQRT_XX_DAG": ("X", True), "SQRT_YY": ("Y", False),
            "SQRT_YY_DAG": ("Y", True), "SQRT_ZZ": ("Z", False), "SQRT_ZZ_DAG": ("Z", True)}
_SWAP_LIKE = {"SWAP", "ISWAP", "ISWAP_DAG", "CXSWAP", "SWAPCX", "CZSWAP", "SWAPCZ"}
_TWO_QUBIT = set(_CONTROLLED) | set(_SQRT_PP) | _SWAP_LIKE
_TO_Z = {"X": "H", "Y": "H_YZ", "Z": None}   # self-inverse, axis -> +Z
_TO_X = {"X": None, "Y": "H_XY", "Z": "H"}   # self-inverse, axis -> +X
_NO_OPS = {"I_ERROR", "II", "II_ERROR"}
# gates that accept a measurement record as their Z-type control: (record is the first operand, Pauli applied)
_FEEDBACK = {"CX": (True, (1, 0)), "CNOT": (True, (1, 0)), "ZCX": (True, (1, 0)),
             "CY": (True, (1, 1)), "ZCY": (True, (1, 1)),
             "CZ": (True, (0, 1)), "ZCZ": (True, (0, 1)),
             "XCZ": (False, (1, 0)), "YCZ": (False, (1, 1))}
_NOISE_1 = {"X_ERROR": (1, 0), "Z_ERROR": (0, 1), "Y_ERROR": (1, 1)}
_IGNORED = {"TICK", "QUBIT_COORDS", "SHIFT_COORDS"}
_MEASURE = {"M": "Z", "MZ": "Z", "MX": "X", "MY": "Y", "MR": "Z", "MRZ": "Z", "MRX": "X", "MRY": "Y"}
_RESET = {"R": "Z", "RZ": "Z", "RX": "X", "RY": "Y"}


@dataclass
class _Analysis:
    channel_probs: list = field(default_factory=list)
    num_e: int = 0
    rec_sets: list = field(default_factory=list)    # error bitmask of every measurement record
    rec_vals: list = field(default_factory=list)    # noiseless outcome of every record
    rec_syms: list = field(default_factory=list)    # random symbols of every record (0: deterministic)
    detectors: list = field(default_factory=list)   # (error bitmask, flip)
    observables: dict = field(default_factory=dict)  # index -> [error bitmask, flip]


class _Sim:
    def __init__(self, n_qubits: int):
        self.n = n_qubits + 1  # one auxiliary qubit for Pauli-product measurements
        self.aux = n_qubits
        self.tab = _Tableau(self.n)
        self.fx = [0] * self.n  # X component of the error frame, as bitmask over e
        self.fz = [0] * self.n
        self.out = _Analysis()
        self.corr_probs: list[float] = []

    # primitives ---------------------------------------------------------------
    def _h(self, q):
        self.tab.h(q)
        self.fx[q], self.fz[q] = self.fz[q], self.fx[q]

    def _s(self, q):
        self.tab.s(q)
        self.fz[q] ^= self.fx[q]

    def _cx(self, c, t):
        self.tab.cx(c, t)
        self.fx[t] ^= self.fx[c]
        self.fz[c] ^= self.fz[t]

    def gate1(self, name, q):
        for step in _ONE_QUBIT[name]:
            if step[0] == "H":
                self._h(q)
            elif step[0] == "S":
                self._s(q)
            else:
                self.tab.pauli(q, step[1], step[2])

    def _basis(self, table, axis, q):
        if table[axis]:
            self.gate1(table[axis], q)

    def _swap(self, a, b):
        self._cx(a, b); self._cx(b, a); self._cx(a, b)

    def _sqrt_zz(self, a, b, dag):  # diag(1, i, i, 1) = CZ (S x S); the inverse for dag
        s = "S_DAG" if dag else "S"
        self.gate1(s, a); self.gate1(s, b)
        self._h(b); self._cx(a, b); self._h(b)

    def gate2(self, name, a, b):
        if a == b:
            raise ValueError(f"{name} {a} {b}: the two targets must differ")
        if name in _CONTROLLED:  # change the control axis to Z and the target axis to X, CX, undo
            ca, tb = _CONTROLLED[name]
            self._basis(_TO_Z, ca, a); self._basis(_TO_X, tb, b)
            self._cx(a, b)
            self._basis(_TO_X, tb, b); self._basis(_TO_Z, ca, a)
        elif name in _SQRT_PP:
            axis, dag = _SQRT_PP[name]
            self._basis(_TO_Z, axis, a); self._basis(_TO_Z, axis, b)
            self._sqrt_zz(a, b, dag)
            self._basis(_TO_Z, axis, b); self._basis(_TO_Z, axis, a)
        elif name == "SWAP":
            self._swap(a, b)
        elif name in ("ISWAP", "ISWAP_DAG"):  # SWAP times sqrt(ZZ)
            self._sqrt_zz(a, b, name == "ISWAP_DAG"); self._swap(a, b)
        elif name == "CXSWAP":
            self._cx(a, b); self._swap(a, b)
        elif name == "SWAPCX":
            self._swap(a, b); self._cx(a, b)
        else:  # CZSWAP == SWAPCZ
            self._h(b); self._cx(a, b); self._h(b); self._swap(a, b)

    def feedback(self, rec_index, q, px, pz):
        """Pauli X^px Z^pz on qubit q if measurement record rec_index is 1 (``CX rec[-1] 3``).  The record
        is const XOR <S, r> XOR <E, e>: the constant acts on the tableau's signs, the random symbols go
        into the signs of the rows that anticommute, the error part into the frame."""
        out = self.out
        if out.rec_vals[rec_index]:
            self.tab.pauli(q, px, pz)
        self.tab.pauli_if(q, out.rec_syms[rec_index], px, pz)
        if px:
            self.fx[q] ^= out.rec_sets[rec_index]
        if pz:
            self.fz[q] ^= out.rec_sets[rec_index]

    def spp(self, paulis, dag=False):
        """Phase the -1 eigenspace of a Pauli product by i (-i for dag): parity of the rotated qubits
        into the last one, S there, undo (instructions.py:966-982)."""
        qs = [q for _, q in paulis]
        if len(set(qs)) != len(qs):
            raise ValueError("SPP: a qubit appears twice in one Pauli product")
        for kind, q in paulis:
            self._basis(_TO_Z, kind, q)
        for q in qs[:-1]:
            self._cx(q, qs[-1])
        self.gate1("S_DAG" if dag else "S", qs[-1])
        for q in reversed(qs[:-1]):
            self._cx(q, qs[-1])
        for kind, q in paulis:
            self._basis(_TO_Z, kind, q)

    # noise ----------------------------------------------------------------------
    def _new_bits(self, k: int) -> list[int]:
        first = self.out.num_e
        if self.corr_probs and first + k > self._shift:
            self._relocate_chain(2 * (first + k) + 64)  # numbered bits must stay below the open chain's bits
        self.out.num_e += k
        return [1 << (first + i) for i in range(k)]

    def error1(self, q, x, z, p):
        self.out.channel_probs.append(error_probs(p))
        (b,) = self._new_bits(1)
        if x:
            self.fx[q] ^= b
        if z:
            self.fz[q] ^= b

    def pauli_channel_1(self, q, px, py, pz):
        self.out.channel_probs.append(pauli_channel_1_probs(px, py, pz))
        bz, bx = self._new_bits(2)  # Z component first (instructions.py:631-638)
        self.fz[q] ^= bz
        self.fx[q] ^= bx

    def pauli_channel_2(self, qi, qj, p15):
        self.out.channel_probs.append(pauli_channel_2_probs(*p15))
        bzi, bxi, bzj, bxj = self._new_bits(4)
        self.fz[qi] ^= bzi; self.fx[qi] ^= bxi
        self.fz[qj] ^= bzj; self.fx[qj] ^= bxj

    def heralded_pauli_channel_1(self, q, pi, px, py, pz):
        """Herald bit into the measurement record, then the Z and X components (instructions.py:726-757)."""
        self.out.channel_probs.append(heralded_pauli_channel_1_probs(pi, px, py, pz))
        bh, bz, bx = self._new_bits(3)
        self.out.rec_sets.append(bh)
        self.out.rec_vals.append(0)
        self.out.rec_syms.append(0)
        self.fz[q] ^= bz
        self.fx[q] ^= bx

    # CORRELATED_ERROR / ELSE_CORRELATED_ERROR chains (instructions.py:759-816): the chain's bits get
    # their error indices, and its table is appended, only when the chain is closed by the next
    # CORRELATED_ERROR or by the end of the circuit - channels seen in between come first.
    # Until then they live at bit positions _shift, _shift + 1, ... - above every numbered error bit; _shift
    # grows (the masks are rewritten) whenever the numbered bits would reach it, so the two ranges never meet.
    _shift = 64

    def _remap_masks(self, fn):
        out = self.out
        self.fx = [fn(m) for m in self.fx]
        self.fz = [fn(m) for m in self.fz]
        out.rec_sets = [fn(m) for m in out.rec_sets]
        out.detectors = [(fn(s), v) for s, v in out.detectors]
        for cur in out.observables.values():
            cur[0] = fn(cur[0])

    def _relocate_chain(self, new_shift):
        old, low = self._shift, (1 << self._shift) - 1
        self._remap_masks(lambda m: (m & low) | ((m >> old) << new_shift))
        self._shift = new_shift

    def correlated_error(self, paulis, p):
        if not self.corr_probs and self.out.num_e > self._shift:
            self._shift = 2 * self.out.num_e + 64  # no chain bit exists yet: nothing to move
        bit = 1 << (self._shift + len(self.corr_probs))
        for kind, q in paulis:
            if kind in ("X", "Y"):
                self.fx[q] ^= bit
            if kind in ("Z", "Y"):
                self.fz[q] ^= bit
        self.corr_probs.append(p)

    def finalize_correlated(self):
        k = len(self.corr_probs)
        if k == 0:
            return
        out = self.out
        out.channel_probs.append(correlated_error_probs(self.corr_probs))
        base, shift, low = out.num_e, self._shift, (1 << self._shift) - 1
        self._remap_masks(lambda m: (m & low) | ((m >> shift) << base))
        out.num_e += k
        self.corr_probs = []

    # measurement / reset -----------------------------------------------------------
    def _basis_in(self, q, basis):
        if basis == "X":
            self.gate1("H", q)
        elif basis == "Y":
            self.gate1("H_YZ", q)

    def measure(self, q, basis="Z", p=0.0, invert=False, reset=False):
        self._basis_in(q, basis)
        flips = self.fx[q]
        if p > 0:  # the outcome flips, the state does not (X before and after, instructions.py:818-840)
            self.out.channel_probs.append(error_probs(p))
            (b,) = self._new_bits(1)
            flips ^= b
        val, sym = self.tab.measure_z(q)
        self.out.rec_sets.append(flips)
        self.out.rec_vals.append(val ^ (1 if invert else 0))
        self.out.rec_syms.append(sym)
        if reset:
            self._reset_z(q)
        self._basis_in(q, basis)  # H and H_YZ are self-inverse

    def _reset_z(self, q):
        val, sym = self.tab.measure_z(q)
        if val:
            self.tab.pauli(q, 1, 0)
        self.tab.pauli_if(q, sym)
        self.fx[q] = 0
        self.fz[q] = 0

    def reset(self, q, basis="Z"):
        self._reset_z(q)
        self._basis_in(q, basis)

    def mpp(self, paulis, p=0.0, invert=False):
        """One Pauli product through the auxiliary qubit (instructions.py:874-909)."""
        a = self.aux
        self._reset_z(a)
        self._h(a)
        for kind, q in paulis:
            self.gate2({"X": "CX", "Z": "CZ", "Y": "CY"}[kind], a, q)
        self._h(a)
        self.measure(a, "Z", p=p, invert=invert)


class CliffordCircuit:
    """A Clifford + Pauli-noise circuit in Stim's text format (the subset listed in the module
    docstring), compiled to the same `(program, channel_probs, error_transform)` triple the
    reference produces for such circuits."""

    def __init__(self, text: str = ""):
        self.text = text
        self.instructions = _parse(text)
        self._compiled = None

    @classmethod
    def from_file(cls, filename: str) -> "CliffordCircuit":
        """Mirror of ``Circuit.from_file`` (src/tsim/circuit.py:292-310)."""
        with open(filename) as fh:
            return cls(fh.read())

    def __str__(self) -> str:
        return self.text

    def __len__(self) -> int:
        return len(self.instructions)  # REPEAT blocks flattened

    @property
    def is_clifford(self) -> bool:
        return True  # anything else is rejected by analyze()

    @property
    def num_qubits(self) -> int:
        return self._qubit_count()

    @property
    def num_measurements(self) -> int:
        return len(self.analyze().rec_sets)

    def without_noise(self) -> "CliffordCircuit":
        """The circuit with every noise instruction dropped and measurement flip arguments removed
        (``Circuit.without_noise``, circuit.py:546-548).  Heralded channels still record their herald."""
        noise = set(_NOISE_1) | {"DEPOLARIZE1", "DEPOLARIZE2", "PAULI_CHANNEL_1", "PAULI_CHANNEL_2", "E",
                                 "CORRELATED_ERROR", "ELSE_CORRELATED_ERROR"}
        lines = []
        for ins in self.instructions:
            if ins.name in noise:
                continue
            if ins.name in ("HERALDED_ERASE", "HERALDED_PAULI_CHANNEL_1"):
                lines.append("MPAD " + " ".join("0" for _ in ins.targets))
                continue
            keep_args = ins.name in ("DETECTOR", "OBSERVABLE_INCLUDE", "QUBIT_COORDS", "SHIFT_COORDS")
            arg = "(" + ", ".join(repr(a) for a in ins.args) + ")" if (ins.args and keep_args) else ""
            lines.append(f"{ins.name}{arg} {' '.join(ins.targets)}".rstrip())
        return CliffordCircuit("\n".join(lines))

    # -- analysis -------------------------------------------------------------------
    def _qubit_count(self) -> int:
        hi = -1
        for ins in self.instructions:
            if ins.name in ("DETECTOR", "OBSERVABLE_INCLUDE") or ins.name in _IGNORED:
                continue
            for t in ins.targets:
                for tok in t.split("*"):
                    m = re.fullmatch(r"!?[XYZxyz]?(\d+)", tok)
                    if m:
                        hi = max(hi, int(m.group(1)))
        return hi + 1

    def analyze(self) -> _Analysis:
        sim = _Sim(max(1, self._qubit_count()))
        out = sim.out

        def rec(tok: str) -> int:
            m = re.fullmatch(r"rec\[-(\d+)\]", tok)
            if not m:
                raise ValueError(f"expected rec[-k], got {tok!r}")
            k = int(m.group(1))
            if k < 1 or k > len(out.rec_sets):
                raise ValueError(f"{tok} reaches before the first measurement")
            return len(out.rec_sets) - k

        for ins in self.instructions:
            name, args, tg = ins.name, ins.args, ins.targets
            if name in _IGNORED or name in _NO_OPS:
                continue
            if any(t.startswith("sweep[") for t in tg):
                raise NotImplementedError(f"{name} with sweep[] targets is not supported by the Clifford front-end")
            if any(t.startswith("rec[") for t in tg) and name not in ("DETECTOR", "OBSERVABLE_INCLUDE"):
                # classically controlled Pauli: the record is the Z-type operand (core/instructions.py cnot/cy/cz/xcz)
                if name not in _FEEDBACK:
                    raise NotImplementedError(f"{name} cannot take a measurement-record target")
                if len(tg) % 2:
                    raise ValueError(f"{name} needs an even number of targets")
                for i in range(0, len(tg), 2):
                    a, b = tg[i], tg[i + 1]
                    ra, rb = a.startswith("rec["), b.startswith("rec[")
                    if not (ra or rb):
                        sim.gate2(name, int(a), int(b))
                        continue
                    ctrl_first, (px, pz) = _FEEDBACK[name]
                    if ra and rb or (ra != ctrl_first and name not in ("CZ", "ZCZ")):
                        raise ValueError(f"{name} {a} {b}: the measurement record must be the Z-controlled operand")
                    sim.feedback(rec(a if ra else b), int(b if ra else a), px, pz)
                continue
            if name in _ONE_QUBIT:
                for t in tg:
                    sim.gate1(name, int(t))
            elif name in _TWO_QUBIT:
                if len(tg) % 2:
                    raise ValueError(f"{name} needs an even number of targets")
                for i in range(0, len(tg), 2):
                    sim.gate2(name, int(tg[i]), int(tg[i + 1]))
            elif name in _NOISE_1:
                x, z = _NOISE_1[name]
                for t in tg:
                    sim.error1(int(t), x, z, args[0])
            elif name == "DEPOLARIZE1":
                for t in tg:
                    sim.pauli_channel_1(int(t), args[0] / 3, args[0] / 3, args[0] / 3)
            elif name == "PAULI_CHANNEL_1":
                for t in tg:
                    sim.pauli_channel_1(int(t), *args)
            elif name in ("DEPOLARIZE2", "PAULI_CHANNEL_2"):
                p15 = (args[0] / 15,) * 15 if name == "DEPOLARIZE2" else args
                for i in range(0, len(tg), 2):
                    sim.pauli_channel_2(int(tg[i]), int(tg[i + 1]), p15)
            elif name in _MEASURE:
                for t in tg:
                    inv = t.startswith("!")
                    sim.measure(int(t.lstrip("!")), _MEASURE[name], p=args[0] if args else 0.0, invert=inv,
                                reset=name.startswith("MR"))
            elif name in _RESET:
                for t in tg:
                    sim.reset(int(t), _RESET[name])
            elif name in ("MXX", "MYY", "MZZ"):
                for i in range(0, len(tg), 2):
                    inv = tg[i].startswith("!") ^ tg[i + 1].startswith("!")
                    sim.mpp([(name[1], int(tg[i].lstrip("!"))), (name[2], int(tg[i + 1].lstrip("!")))],
                            p=args[0] if args else 0.0, invert=inv)
            elif name == "MPAD":  # a fixed bit in the measurement record (instructions.py:1040-1053)
                for t in tg:
                    flips = 0
                    if args and args[0] > 0:
                        out.channel_probs.append(error_probs(args[0]))
                        (flips,) = sim._new_bits(1)
                    out.rec_sets.append(flips)
                    out.rec_vals.append(int(t) & 1)
                    out.rec_syms.append(0)
            elif name == "MPP":
                for t in tg:
                    inv = t.startswith("!")
                    paulis = [(tok[0].upper(), int(tok[1:])) for tok in t.lstrip("!").split("*")]
                    sim.mpp(paulis, p=args[0] if args else 0.0, invert=inv)
            elif name in ("SPP", "SPP_DAG"):
                for t in tg:
                    paulis = [(tok[0].upper(), int(tok[1:])) for tok in t.lstrip("!").split("*")]
                    sim.spp(paulis, dag=(name == "SPP_DAG") ^ t.startswith("!"))
            elif name in ("HERALDED_ERASE", "HERALDED_PAULI_CHANNEL_1"):
                pr = (args[0] / 4,) * 4 if name == "HERALDED_ERASE" else args
                for t in tg:
                    sim.heralded_pauli_channel_1(int(t), *pr)
            elif name in ("E", "CORRELATED_ERROR", "ELSE_CORRELATED_ERROR"):
                if name != "ELSE_CORRELATED_ERROR":
                    sim.finalize_correlated()
                sim.correlated_error([(t[0].upper(), int(t[1:])) for t in tg], args[0])
            elif name == "DETECTOR":
                s, v, y = 0, 0, 0
                for t in tg:
                    s ^= out.rec_sets[rec(t)]
                    v ^= out.rec_vals[rec(t)]
                    y ^= out.rec_syms[rec(t)]
                if y:
                    raise ValueError(f"DETECTOR {' '.join(tg)} (detector {len(out.detectors)}) is not deterministic: "
                                     "its measurements depend on random outcomes that do not cancel")
                out.detectors.append((s, v))
            elif name == "OBSERVABLE_INCLUDE":
                idx = int(args[0]) if args else 0
                cur = out.observables.setdefault(idx, [0, 0, 0])
                for t in tg:
                    cur[0] ^= out.rec_sets[rec(t)]
                    cur[1] ^= out.rec_vals[rec(t)]
                    cur[2] ^= out.rec_syms[rec(t)]
            else:
                raise NotImplementedError(f"instruction {name} is not supported by the Clifford front-end")
        sim.finalize_correlated()
        return sim.out

    # -- program assembly -------------------------------------------------------------------
    def compile(self):
        """``(program, channel_probs, error_transform)`` for detector sampling."""
        if self._compiled is not None:
            return self._compiled
        an = self.analyze()
        for k, (_, _, y) in an.observables.items():
            if y:
                raise ValueError(f"OBSERVABLE {k} is not deterministic: its random outcomes do not cancel")
        outputs = list(an.detectors) + [tuple(an.observables[k][:2]) for k in sorted(an.observables)]
        n_det, n_out = len(an.detectors), len(outputs)
        # rows of the error matrix: outputs that carry at least one error bit, in output order
        rows = [(i, s) for i, (s, _) in enumerate(outputs) if s]
        basis_idx, combos = find_basis([s for _, s in rows])
        num_f = len(basis_idx)
        need_zero_col = any(not s for s, _ in outputs)
        error_transform = np.zeros((num_f + (1 if need_zero_col else 0), an.num_e), dtype=np.uint8)
        for pos, bi in enumerate(basis_idx):
            s = rows[bi][1]
            for e in range(an.num_e):
                if (s >> e) & 1:
                    error_transform[pos, e] = 1
        combo_of = {i: combos[j] for j, (i, _) in enumerate(rows)}
        direct, components = [], []
        for i, (s, flip) in enumerate(outputs):
            c = combo_of.get(i, 0)
            if not s:
                direct.append((i, num_f, bool(flip)))  # the always-zero column
            elif c & (c - 1) == 0:
                direct.append((i, c.bit_length() - 1, bool(flip)))
            else:  # XOR of several basis bits: deterministic one-output component
                fsel = [b for b in range(num_f) if (c >> b) & 1]
                F = len(fsel)
                lv0 = scalar_graphs_from_terms(F, [dict(floatfactor=(1, 0, 0, 0), power2=0)])
                lv1 = scalar_graphs_from_terms(F + 1, [dict(floatfactor=(1, 0, 0, 0), power2=-1,
                                                            A=[(4 * int(flip), list(range(F + 1)))])])
                components.append(CompiledComponent((i,), np.asarray(fsel, np.int32), (lv0, lv1)))
        program = make_program(components, direct, n_out, n_det)
        self._compiled = (program, list(an.channel_probs), error_transform)
        return self._compiled

    def compile_measurements(self):
        """``(program, channel_probs, error_transform)`` for measurement sampling (every record is an
        output; ``Circuit.compile_sampler``, src/tsim/circuit.py:812-834)."""
        if getattr(self, "_compiled_m", None) is not None:
            return self._compiled_m
        an = self.analyze()
        n_out = len(an.rec_sets)
        rows = [(i, s) for i, s in enumerate(an.rec_sets) if s]
        basis_idx, combos = find_basis([s for _, s in rows])
        num_f = len(basis_idx)
        error_transform = np.zeros((num_f, an.num_e), dtype=np.uint8)
        for pos, bi in enumerate(basis_idx):
            for e in range(an.num_e):
                if (rows[bi][1] >> e) & 1:
                    error_transform[pos, e] = 1
        fcombo = [0] * n_out
        for j, (i, _) in enumerate(rows):
            fcombo[i] = combos[j]
        # direct entries: no random bit and exactly one f bit
        direct, rest = [], []
        for i in range(n_out):
            c = fcombo[i]
            if an.rec_syms[i] == 0 and c and c & (c - 1) == 0:
                direct.append((i, c.bit_length() - 1, bool(an.rec_vals[i])))
            else:
                rest.append(i)
        # components: outputs connected through shared random bits
        parent = {i: i for i in rest}

        def find(i):
            while parent[i] != i:
                parent[i] = parent[parent[i]]
                i = parent[i]
            return i

        owner: dict[int, int] = {}
        for i in rest:
            y, b = an.rec_syms[i], 0
            while y:
                if y & 1:
                    if b in owner:
                        parent[find(i)] = find(owner[b])
                    else:
                        owner[b] = i
                y >>= 1
                b += 1
        groups: dict[int, list[int]] = {}
        for i in rest:
            groups.setdefault(find(i), []).append(i)
        components = []
        for outs in sorted(groups.values(), key=lambda g: (len(g), g[0])):
            outs = sorted(outs)
            fsel = sorted({b for i in outs for b in range(num_f) if (fcombo[i] >> b) & 1})
            fpos = {b: k for k, b in enumerate(fsel)}
            F = len(fsel)
            levels = [scalar_graphs_from_terms(F, [dict(floatfactor=(1, 0, 0, 0), power2=0)])]
            sym_basis: list[tuple[int, int, int]] = []  # (reduced symbol vector, pivot, members as bitmask over positions)
            deltas: list[tuple[int, list[int]]] = []     # NodePhases terms accumulated so far
            for k, i in enumerate(outs):
                v, members = an.rec_syms[i], 1 << k
                for rv, pivot, mem in sym_basis:
                    if v & pivot:
                        v ^= rv
                        members ^= mem
                if v:
                    sym_basis.append((v, v & -v, members))  # free output: marginal 1/2
                else:
                    # XOR of the member outputs is const XOR <f>: a delta over those output bits and f bits
                    const, fmask = 0, 0
                    params = []
                    for q, io in enumerate(outs[: k + 1]):
                        if (members >> q) & 1:
                            const ^= an.rec_vals[io]
                            fmask ^= fcombo[io]
                            params.append(F + q)
                    params = [fpos[b] for b in range(num_f) if (fmask >> b) & 1] + params
                    deltas.append((4 * const, params))
                levels.append(scalar_graphs_from_terms(
                    F + k + 1, [dict(floatfactor=(1, 0, 0, 0), power2=-(k + 1), A=list(deltas))]))
            components.append(CompiledComponent(tuple(outs), np.asarray(fsel, np.int32), tuple(levels)))
        program = make_program(components, direct, n_out, 0)
        self._compiled_m = (program, list(an.channel_probs), error_transform)
        return self._compiled_m

    def compile_sampler(self, *, seed: int | None = None, device: int = 0, noise: str = "host", mode: str = "auto"):
        """Mirror of ``Circuit.compile_sampler`` (src/tsim/circuit.py:812-834)."""
        from .sampler import CompiledMeasurementSampler

        program, channel_probs, error_transform = self.compile_measurements()
        return CompiledMeasurementSampler(program, channel_probs=channel_probs, error_transform=error_transform,
                                          seed=seed, device=device, noise=noise, mode=mode)

    def compile_detector_sampler(self, *, seed: int | None = None, device: int = 0, noise: str = "host",
                                 mode: str = "auto"):
        """Mirror of ``Circuit.compile_detector_sampler`` (src/tsim/circuit.py:836-867)."""
        from .sampler import CompiledDetectorSampler

        program, channel_probs, error_transform = self.compile()
        return CompiledDetectorSampler(program, channel_probs=channel_probs, error_transform=error_transform,
                                       seed=seed, device=device, noise=noise, mode=mode)

    @property
    def num_detectors(self) -> int:
        return int(self.compile()[0].num_detectors)

    @property
    def num_observables(self) -> int:
        p = self.compile()[0]
        return int(p.num_outputs - p.num_detectors)
