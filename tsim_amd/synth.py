"""Synthetic and hand-built compiled programs.

The reference's compile pipeline (stim parsing, ZX rewriting with ``pyzx_param``,
stabiliser decomposition - /root/reference/src/tsim/compile/pipeline.py,
stabrank.py, core/graph.py) is out of scope, so programs are
either

* **hand-built**: tiny programs whose amplitudes are written down by hand for
  the circuits of the reference's seeded known-answer tests
  (test/unit/test_sampler.py:223-233, test/integration/test_sampler_circuits.py:10-109),
* **synthetic**: seeded random programs with the *shape* of the BASELINE.json
  configurations (C1..C5), or
* **imported** from a machine that has tsim installed (``program.load_npz``).
"""

from __future__ import annotations

import numpy as np

from .program import (
    CompiledComponent,
    CompiledProgram,
    empty_scalar_graphs,
    make_program,
    scalar_graphs_from_terms,
)

# --------------------------------------------------------------------------
# hand-built programs for the reference's seeded KATs
# --------------------------------------------------------------------------


def _const_level(n_params: int, floatfactor=(1, 0, 0, 0), power2=0, **terms) -> object:
    g = dict(floatfactor=floatfactor, power2=power2)
    g.update(terms)
    return scalar_graphs_from_terms(n_params, [g])


def single_output_component(output_index: int, ff1=(1, 0, 0, 0), power2_1=-1, zero=False, one=False,
                            level1=None) -> CompiledComponent:
    """One output, no noise: level 0 amplitude 1, level 1 amplitude ``ff1 * 2^power2_1``.

    ``zero=True`` / ``one=True`` give a deterministic outcome through a delta term
    ``(1 +- (-1)^m)/2`` (a NodePhases factor with phase 0 / 4 on the output bit), so that the
    marginals still sum to the normalisation (sampler.py:71-72).  ``level1`` overrides the
    level-1 graph list.
    """
    lv0 = _const_level(0)
    if level1 is not None:
        lv1 = scalar_graphs_from_terms(1, level1)
    elif zero or one:
        lv1 = _const_level(1, power2=-1, A=[(4 if one else 0, [0])])
    else:
        lv1 = _const_level(1, floatfactor=ff1, power2=power2_1)
    return CompiledComponent((output_index,), np.zeros(0, np.int32), (lv0, lv1))


def kat_h_m() -> CompiledProgram:
    """``H 0; M 0`` - P(1) = 1/2 (test_sampler.py:223-233)."""
    return make_program([single_output_component(0)], [], 1, 0)


def kat_t_gate() -> CompiledProgram:
    """``RX 0; S[T] 0; H 0; M 0`` - P(m) = 1/2 + (-1)^m sqrt2/4 (test_sampler_circuits.py:40-49).

    Two stabiliser terms: ``1/2`` and ``sqrt2/4 * w^(4m)`` with sqrt2 = w + conj(w), i.e.
    floatfactor (0,1,0,1) * 2^-2 and a HalfPi term of coefficient 4 on the output bit.
    P(1) = (2 - sqrt2)/4 = sin^2(pi/8): exact sum (2,-1,0,-1) * 2^-2.
    """
    level1 = [dict(power2=-1), dict(floatfactor=(0, 1, 0, 1), power2=-2, B=[(4, [0])])]
    return make_program([single_output_component(0, level1=level1)], [], 1, 0)


def kat_r_gate() -> CompiledProgram:
    """``RX 0; RX 0; M 0; RX 0; M 0; R 0; M 0`` (test_sampler_circuits.py:90-109).

    Three independent single-output components with P(1) = 1/2, 1/2, 0.
    """
    comps = [
        single_output_component(0),
        single_output_component(1),
        single_output_component(2, zero=True),
    ]
    return make_program(comps, [], 3, 0)


def kat_bell() -> CompiledProgram:
    """``R 0 1; H 0; CNOT 0 1; M 0 1`` (test_sampler_circuits.py:10-22).

    One two-output component: P(m0=1) = 1/2, P(m0, m1) = 1/4 (1 + (-1)^(m0+m1)).
    """
    lv0 = _const_level(0)
    lv1 = _const_level(1, power2=-1)
    lv2 = _const_level(2, power2=-2, A=[(0, [0, 1])])
    comp = CompiledComponent((0, 1), np.zeros(0, np.int32), (lv0, lv1, lv2))
    return make_program([comp], [], 2, 0)


def kat_x_error_detector() -> CompiledProgram:
    """Bell pair + ``X_ERROR(0.3) 0`` + ``DETECTOR rec[-1] rec[-2]`` (test_sampler_circuits.py:25-37).

    The detector is the single error bit: fully direct, one f variable.
    """
    return make_program([], [(0, 0, False)], 1, 1)


def noisy_t_component(output_index: int, f_index: int) -> CompiledComponent:
    """``RX; T; H; M`` with an X error before M, controlled by one f bit.

    P(m | f) = 1/2 + (-1)^(m+f) sqrt2/4, written as two stabiliser terms:
    ``1/2`` and ``(omega + conj omega)/4 * omega^(4 (m xor f))`` (a HalfPi term).
    A physically consistent 2-graph program, used by statistical tests.
    """
    lv0 = scalar_graphs_from_terms(1, [dict()])
    lv1 = scalar_graphs_from_terms(
        2,
        [
            dict(power2=-1),
            dict(floatfactor=(0, 1, 0, 1), power2=-2, B=[(4, [0, 1])]),
        ],
    )
    return CompiledComponent((output_index,), np.asarray([f_index], np.int32), (lv0, lv1))


# --------------------------------------------------------------------------
# seeded synthetic programs
# --------------------------------------------------------------------------


def _rand_bits(rng: np.random.Generator, P: int, density: float) -> list[int]:
    if P == 0:
        return []
    m = rng.random(P) < density
    if not m.any():
        m[rng.integers(0, P)] = True
    return np.flatnonzero(m).tolist()


def synth_level(
    rng: np.random.Generator,
    n_params: int,
    G: int,
    *,
    ta=(4, 16),
    tb=(4, 24),
    tc=(8, 40),
    td=(0, 4),
    density: float = 0.3,
    approx: bool = False,
    zero_phase_fraction: float = 0.1,
):
    """One level with ``G`` random graphs; term counts uniform in the given ranges."""
    graphs = []
    for _ in range(G):
        nA = int(rng.integers(ta[0], ta[1] + 1))
        nB = int(rng.integers(tb[0], tb[1] + 1))
        nC = int(rng.integers(tc[0], tc[1] + 1))
        nD = int(rng.integers(td[0], td[1] + 1))
        A = []
        for _t in range(nA):
            if rng.random() < zero_phase_fraction:
                ph = int(rng.choice([0, 4]))
            else:
                ph = int(rng.choice([1, 2, 3, 5, 6, 7]))
            A.append((ph, _rand_bits(rng, n_params, density)))
        B = [(int(rng.choice([2, 4, 6])), _rand_bits(rng, n_params, density)) for _t in range(nB)]
        C = [
            (
                int(rng.integers(0, 2)),
                _rand_bits(rng, n_params, density),
                int(rng.integers(0, 2)),
                _rand_bits(rng, n_params, density),
            )
            for _t in range(nC)
        ]
        D = [
            (
                int(rng.integers(0, 8)),
                _rand_bits(rng, n_params, density),
                int(rng.integers(0, 8)),
                _rand_bits(rng, n_params, density),
            )
            for _t in range(nD)
        ]
        ff = rng.integers(-2, 3, size=4)
        if not ff.any():
            ff[0] = 1
        g = dict(
            A=A,
            B=B,
            C=C,
            D=D,
            phase=int(rng.integers(0, 8)),
            floatfactor=tuple(int(v) for v in ff),
            power2=int(rng.integers(-6, 1)),
        )
        if approx:
            g["approx"] = complex(np.exp(2j * np.pi * rng.random()) * (0.5 + rng.random()))
        graphs.append(g)
    return scalar_graphs_from_terms(n_params, graphs)


def split_graphs(total: int, parts: int) -> list[int]:
    """Split ``total`` graphs over ``parts`` levels, growing with the level index."""
    w = np.arange(1, parts + 1, dtype=np.float64)
    g = np.maximum(1, np.floor(total * w / w.sum()).astype(int))
    g[-1] += total - int(g.sum())
    return [int(v) for v in g]


def synth_program(
    *,
    num_f: int,
    n_direct: int,
    components: list[dict],
    seed: int = 42,
    num_detectors: int | None = None,
    shuffle_outputs: bool = False,
    direct_flip_fraction: float = 0.0,
    identity_direct: bool = True,
) -> CompiledProgram:
    """Random program.  ``components`` entries: ``dict(n=, F=, G=[per level], **synth_level kwargs)``."""
    rng = np.random.default_rng(seed)
    n_comp_out = sum(c["n"] for c in components)
    num_outputs = n_direct + n_comp_out
    out_ids = np.arange(num_outputs)
    if shuffle_outputs:
        out_ids = rng.permutation(num_outputs)
    if identity_direct:
        dfi = np.arange(n_direct)
    else:
        dfi = rng.choice(num_f, size=n_direct, replace=False) if n_direct else np.zeros(0, int)
    flips = rng.random(n_direct) < direct_flip_fraction
    direct = [(int(out_ids[j]), int(dfi[j]), bool(flips[j])) for j in range(n_direct)]
    comps = []
    pos = n_direct
    for c in components:
        n, F = int(c["n"]), int(c["F"])
        Gs = list(c["G"])
        if len(Gs) != n + 1:
            raise ValueError("need n+1 graph counts per component")
        kw = {k: v for k, v in c.items() if k not in ("n", "F", "G")}
        fsel = np.sort(rng.choice(num_f, size=F, replace=False)).astype(np.int32)
        levels = tuple(synth_level(rng, F + k, Gs[k], **kw) for k in range(n + 1))
        comps.append(CompiledComponent(tuple(int(i) for i in out_ids[pos : pos + n]), fsel, levels))
        pos += n
    nd = num_detectors if num_detectors is not None else n_direct
    return make_program(comps, direct, num_outputs, nd)


def synth_f(B: int, num_f: int, p_bit: float, seed: int = 42) -> np.ndarray:
    """Seeded ``uint8[B, num_f]`` batch, every bit fires independently w.p. ``p_bit``."""
    rng = np.random.default_rng(seed)
    return (rng.random((B, num_f)) < p_bit).astype(np.uint8)


# BASELINE.json configurations

CONFIGS = {
    # d=3 rotated surface code, Clifford only: pure direct path
    "C1": dict(name="d=3 surface code, Clifford only (direct path)", num_f=24, n_direct=24, components=[], shots=1000, batch=1000, p_bit=0.008, seed=0),
    # 35-qubit distillation: 15 direct detectors + one 5-output component, sum G = 148
    "C2": dict(
        name="35-qubit distillation shape (SURVEY 8d)",
        num_f=64,
        n_direct=15,
        components=[dict(n=5, F=32, G=[8, 16, 24, 28, 32, 40])],
        shots=1_000_000,
        batch=1 << 17,
        p_bit=0.02,
        seed=42,
    ),
    # 85-qubit distillation: sum G = 147
    "C3": dict(
        name="85-qubit distillation shape (SURVEY 8d)",
        num_f=104,
        n_direct=40,
        components=[dict(n=5, F=48, G=[7, 16, 24, 28, 32, 40])],
        shots=1_000_000,
        batch=1 << 17,
        p_bit=0.02,
        seed=42,
    ),
    # d=3 cultivation: three components, sum G = 1024, T up to 64
    "C4": dict(
        name="d=3 cultivation shape (SURVEY 8d)",
        num_f=64,
        n_direct=10,
        components=[
            dict(n=1, F=12, G=[2, 6], ta=(4, 24), tb=(4, 32), tc=(8, 64), td=(0, 4)),
            dict(n=1, F=12, G=[2, 6], ta=(4, 24), tb=(4, 32), tc=(8, 64), td=(0, 4)),
            dict(n=6, F=40, G=[16, 48, 96, 160, 208, 224, 256], ta=(4, 24), tb=(4, 32), tc=(8, 64), td=(0, 4)),
        ],
        shots=100_000,
        batch=1 << 15,
        p_bit=0.02,
        seed=42,
    ),
    # d=5 surface code + injected T: wide f (W=4), few graphs
    "C5": dict(
        name="d=5 surface code + T shape (SURVEY 8d)",
        num_f=320,
        n_direct=118,
        components=[dict(n=3, F=200, G=[1, 2, 2, 3], density=0.08)],
        shots=1_000_000,
        batch=1 << 17,
        p_bit=0.02,
        seed=42,
    ),
}


def config_program(name: str, *, approx: bool = False, physical: bool = True, live_padding: bool = False) -> tuple[CompiledProgram, dict]:
    """Return ``(program, config_dict)`` for one of ``C1..C5``.

    ``physical=True`` (default): the normalised probability model of :func:`physical_program`;
    ``physical=False``: the unconstrained random program of :func:`synth_program` (nonsense marginals,
    NaN / negative thresholds - the kernels must equal the oracle on those too).
    ``live_padding=True`` (physical programs): the term counts of the published shape are reached with terms that DO
    something - a quadratic phase common to every graph and per-lineage weight factors - instead of neutral pairs that
    the packer's algebra cancels (see :func:`physical_component`)."""
    cfg = CONFIGS[name]
    if physical:
        prog = physical_program(num_f=cfg["num_f"], n_direct=cfg["n_direct"], components=[dict(c) for c in cfg["components"]],
                                seed=cfg["seed"], approx=approx, live_padding=live_padding)
        return prog, dict(cfg, name=cfg["name"] + ", normalised probability model" + (", live padding" if live_padding else ""))
    comps = [dict(c, approx=approx) if approx else dict(c) for c in cfg["components"]]
    prog = synth_program(
        num_f=cfg["num_f"], n_direct=cfg["n_direct"], components=comps, seed=cfg["seed"]
    )
    return prog, cfg


# --------------------------------------------------------------------------
# physically normalised synthetic programs
# --------------------------------------------------------------------------
#
# The random programs above have the published SHAPE but their "amplitudes" are not probabilities:
# p1/prev is often outside [0, 1], so many Bernoulli draws are decided whatever the float value.
# The programs below are genuine probability models with the same shape - a mixture of product
# distributions, written in the reference's own term families - so that for every level i
#
#     amp_i(f, m_<i, 0) + amp_i(f, m_<i, 1) == amp_{i-1}(f, m_<i)        exactly (in Z[w] * 2^k)
#     amp_i >= 0
#
# and therefore every threshold p1/prev lies in [0, 1] and the normalisation check
# (sampler.py:66-72) deviates from 1 only by float32 rounding.
#
# Construction.  A *lineage* is a set of graphs (stabiliser terms) that is extended in lockstep, one
# conditional factor q_j(m_j | f, m_<j) per output, with sum_{m_j} q_j = 1:
#   "T"  NodePhases pair (k, 8-k), k in {1, 3}, on the same parity row r + {m_j}:
#        (1 + w^(k+4p))(1 + w^(-k+4p)) = 2 +- sqrt2 (-1)^p, power2 -= 2      -> (2 +- sqrt2)/4
#   "U"  NodePhases pair (2, 6): (1 + i^..)(1 - i^..) = 2, power2 -= 2         -> 1/2
#   "D"  one NodePhases term of phase 0 / 4: 1 +- (-1)^p, power2 -= 1          -> delta(p = 0 / 1)
#   "S"  the "T" factor split over TWO graphs: 1/2  and  (w + conj w)/4 * w^(4p) (a HalfPi term of
#        coefficient 4 and floatfactor * (0,1,0,1)): the second one alone is signed, their sum is not.
# Members of a lineage differ by what they carry from level 0 on and keep at every level:
#   * a positive weight (floatfactor * 2^power2), optionally f-dependent through "T" factors on f rows;
#   * sign factors s(f) = +-1 on f rows - PiProducts (-1)^(psi phi) and PhasePairs with alpha, beta in
#     {0, 4} (1 + w^a + w^b - w^(a+b) = 2 (-1)^(pa pb), power2 -= 1) - always next to an unsigned
#     member of three times the weight, so that the lineage's sum stays positive.
# Every graph is then padded with *neutral* terms on rows over ALL parameters (HalfPi pairs of
# coefficients (2,6) or (4,4) on one row, duplicated PiProducts, duplicated sign-type PhasePairs with
# power2 -= 2, NodePhases (2,6) pairs with power2 -= 1) up to the term counts of the published shape:
# they change no value but are real work for the oracle and the faithful kernel.


def _zw_mul(x, y):
    """Product in Z[w] on the basis (1, w, i, conj w) (exact_scalar.py:19-39)."""
    a1, b1, c1, d1 = x
    a2, b2, c2, d2 = y
    return (
        a1 * a2 + b1 * d2 - c1 * c2 + d1 * b2,
        a1 * b2 + b1 * a2 + c1 * d2 + d1 * c2,
        a1 * c2 + b1 * b2 + c1 * a2 - d1 * d2,
        a1 * d2 - b1 * c2 - c1 * b2 + d1 * a2,
    )


def _zw_norm(ff, power2):
    """Pull common factors of two out of a floatfactor."""
    ff = tuple(int(v) for v in ff)
    while any(ff) and all(v % 2 == 0 for v in ff):
        ff = tuple(v // 2 for v in ff)
        power2 += 1
    return ff, power2


def _copy_graph(g: dict) -> dict:
    return dict(A=list(g["A"]), B=list(g["B"]), C=list(g["C"]), D=list(g["D"]), phase=g["phase"],
                floatfactor=tuple(g["floatfactor"]), power2=g["power2"], approx=g.get("approx", 1.0))


def _rand_row(rng, lo: int, hi: int, density: float) -> list[int]:
    """Random non-empty subset of the parameter indices [lo, hi)."""
    n = hi - lo
    if n <= 0:
        return []
    m = rng.random(n) < density
    if not m.any():
        m[rng.integers(0, n)] = True
    return (lo + np.flatnonzero(m)).tolist()


def _sign_terms(rng, F: int, density: float, n_c: int, n_d: int):
    """Random f-only sign factors: ``n_c`` PiProducts and ``n_d`` sign-type PhasePairs (+ power shift)."""
    C = [(int(rng.integers(0, 2)), _rand_row(rng, 0, F, density), int(rng.integers(0, 2)), _rand_row(rng, 0, F, density))
         for _ in range(n_c)]
    D = [(int(rng.choice([0, 4])), _rand_row(rng, 0, F, density), int(rng.choice([0, 4])), _rand_row(rng, 0, F, density))
         for _ in range(n_d)]
    return C, D, -n_d


def _pad_neutral(rng, g: dict, P: int, density: float, ta, tb, tc, td) -> dict:
    """Neutral terms up to the shape's term counts (drawn per graph from the given ranges)."""
    g = _copy_graph(g)
    if P == 0:
        return g
    want_a = int(rng.integers(ta[0], ta[1] + 1))
    want_b = int(rng.integers(tb[0], tb[1] + 1))
    want_c = int(rng.integers(tc[0], tc[1] + 1))
    want_d = int(rng.integers(td[0], td[1] + 1))
    while len(g["A"]) + 2 <= want_a:
        row = _rand_row(rng, 0, P, density)
        g["A"] += [(2, row), (6, row)]
        g["power2"] -= 1
    while len(g["B"]) + 2 <= want_b:
        row = _rand_row(rng, 0, P, density)
        g["B"] += [(2, row), (6, row)] if rng.random() < 0.5 else [(4, row), (4, row)]
    while len(g["C"]) + 2 <= want_c:
        t = (int(rng.integers(0, 2)), _rand_row(rng, 0, P, density), int(rng.integers(0, 2)), _rand_row(rng, 0, P, density))
        g["C"] += [t, t]
    while len(g["D"]) + 2 <= want_d:
        t = (int(rng.choice([0, 4])), _rand_row(rng, 0, P, density), int(rng.choice([0, 4])), _rand_row(rng, 0, P, density))
        g["D"] += [t, t]
        g["power2"] -= 2
    # interleave: the families are products, term order is free (rows of one pair need not be adjacent)
    for fam in "ABCD":
        order = rng.permutation(len(g[fam]))
        g[fam] = [g[fam][int(k)] for k in order]
    return g


def physical_component(rng, output_indices, f_selection, Gs, *, density: float = 0.3, ta=(4, 16), tb=(4, 24),
                       tc=(8, 40), td=(0, 4), approx: bool = False, delta_fraction: float = 0.08,
                       signed_fraction: float = 0.35, live_padding: bool = False,
                       shared_delta: float = 0.0) -> CompiledComponent:
    """One normalised component with ``Gs[k]`` graphs at level k (see the block comment above).

    ``live_padding``: the neutral pairs that bring a graph to the published term counts are exact no-ops, and the
    packer's GF(2) algebra removes them (C2: 10 972 rows in, 2 510 kept).  With ``live_padding`` the same counts are
    reached by terms no algebra can remove, the model staying a normalised mixture:
      * a quadratic phase w^q(f) COMMON to every graph of every level - HalfPi rows with coefficients 2/4/6 and
        PiProducts on f-only rows, about tb/2 and tc/2 of them: amp_i -> w^q(f) amp_i for all i, so |amp| and every
        threshold are unchanged as real numbers while each graph's quadratic form has full rank (Dickson pairs) and the
        amplitudes are genuinely complex (the general |z| path);
      * per LINEAGE (identical in all its members, from level 0 on) ta/2 "T" pairs (k, 8 - k) on f rows: positive
        f-dependent weights (2 +- sqrt2)/4 - counted NodePhases rows.

    ``shared_delta``: the fraction of outputs whose conditional factor is ONE delta (1 + (-1)^(row . x + c)) / 2 shared by
    every lineage - a detector that is a deterministic parity of the error bits and the earlier outcomes, the regime of real
    detector components (most detectors of a connected component are fixed by f and a few genuinely random outcomes).
    Without it every output multiplies every graph by a (2 +- sqrt2)-type weight, and beyond ~22 outputs the int32
    coefficients of the reference's exact scalars wrap (normalisation deviation 1: the reference raises)."""
    n, F = len(output_indices), len(f_selection)
    if len(Gs) != n + 1:
        raise ValueError("need n+1 graph counts per component")
    if any(b < a for a, b in zip(Gs, Gs[1:])) or Gs[0] < 1:
        raise ValueError("graph counts must be positive and non-decreasing")
    # ---- level 0: lineages of one unsigned member, or an (unsigned x3, signed x1) pair
    lineages: list[list[dict]] = []
    left = Gs[0]
    while left > 0:
        base = dict(A=[], B=[], C=[], D=[], phase=0, floatfactor=(int(rng.integers(1, 4)), 0, 0, 0),
                    power2=int(rng.integers(-3, 1)), approx=1.0)
        if approx:
            base["approx"] = complex(0.5 + rng.random())  # real positive: mixture weights stay positive
        if F > 0:
            n_t = int(rng.integers(max(1, ta[0] // 2), max(2, ta[1] // 2) + 1)) if live_padding else int(rng.integers(0, 3))
            for _ in range(n_t):  # f-dependent weight
                k = int(rng.choice([1, 3]))
                row = _rand_row(rng, 0, F, density)
                base["A"] += [(k, row), (8 - k, row)]
                base["power2"] -= 2
        if left >= 2 and F > 0 and rng.random() < signed_fraction:
            big, small = _copy_graph(base), _copy_graph(base)
            big["floatfactor"] = tuple(3 * v for v in base["floatfactor"])
            C, D, dp = _sign_terms(rng, F, density, int(rng.integers(max(1, tc[0] // 2), max(2, tc[1] // 2) + 1)),
                                   int(rng.integers(0, max(1, td[1] // 2) + 1)))
            small["C"], small["D"] = C, D
            small["power2"] += dp
            lineages.append([big, small])
            left -= 2
        else:
            lineages.append([base])
            left -= 1
    levels = []
    common_B, common_C = [], []
    if live_padding and F > 0:
        common_B = [(int(rng.choice([2, 4, 6])), _rand_row(rng, 0, F, density)) for _ in range(int(rng.integers(max(1, tb[0] // 2), max(2, tb[1] // 2) + 1)))]
        common_C = [(int(rng.integers(0, 2)), _rand_row(rng, 0, F, density), int(rng.integers(0, 2)), _rand_row(rng, 0, F, density))
                    for _ in range(int(rng.integers(max(1, tc[0] // 2), max(2, tc[1] // 2) + 1)))]

    def emit(k: int):
        if live_padding:
            graphs = []
            for lin in lineages:
                for g in lin:
                    g2 = _copy_graph(g)
                    g2["B"] = list(g2["B"]) + common_B
                    g2["C"] = list(g2["C"]) + common_C
                    for fam in "ABCD":  # term order inside a family is free
                        order = rng.permutation(len(g2[fam]))
                        g2[fam] = [g2[fam][int(q)] for q in order]
                    graphs.append(g2)
        else:
            graphs = [_pad_neutral(rng, g, F + k, density, ta, tb, tc, td) for lin in lineages for g in lin]
        assert len(graphs) == Gs[k]
        return scalar_graphs_from_terms(F + k, graphs)

    levels.append(emit(0))
    # ---- one conditional factor per output
    for j in range(n):
        bit = F + j
        grow = Gs[j + 1] - Gs[j]
        order = rng.permutation(len(lineages))
        split = set()
        if shared_delta > 0.0 and rng.random() < shared_delta:
            row = (_rand_row(rng, 0, bit, density) + [bit]) if bit > 0 else [bit]
            cst = int(rng.choice([0, 4]))
            for lin in lineages:
                for g in lin:
                    g["A"].append((cst, row))
                    g["power2"] -= 1
            while grow > 0:
                lin = lineages[int(rng.integers(0, len(lineages)))]
                src = int(rng.integers(0, len(lin)))
                lin[src]["power2"] -= 1
                lin.insert(src + 1, _copy_graph(lin[src]))
                grow -= 1
            levels.append(emit(j + 1))
            continue
        for li in order:  # lineages that take the two-graph form of the factor
            s = len(lineages[int(li)])
            if s <= grow and rng.random() < 0.8:
                split.add(int(li))
                grow -= s
        for li, lin in enumerate(lineages):
            row = _rand_row(rng, 0, bit, density) + [bit] if bit > 0 else [bit]
            if bit > 0 and rng.random() < 0.15:
                row = [bit]  # an output that ignores its context
            if li in split:
                sign = int(rng.choice([0, 4]))
                new = []
                for g in lin:
                    g1, g2 = _copy_graph(g), _copy_graph(g)
                    g1["power2"] -= 1
                    g2["B"].append((4, row))
                    g2["phase"] = (g2["phase"] + sign) % 8
                    g2["floatfactor"], g2["power2"] = _zw_norm(_zw_mul(g2["floatfactor"], (0, 1, 0, 1)), g2["power2"] - 2)
                    new += [g1, g2]
                lin[:] = new
                continue
            r = rng.random()
            for g in lin:
                if r < delta_fraction:
                    g["A"].append((0 if r < delta_fraction / 2 else 4, row))
                    g["power2"] -= 1
                elif r < delta_fraction + 0.12:
                    g["A"] += [(2, row), (6, row)]
                    g["power2"] -= 2
                else:
                    k = 1 if r < 0.6 else 3
                    g["A"] += [(k, row), (8 - k, row)]
                    g["power2"] -= 2
        while grow > 0:  # duplicates: w = w/2 + w/2 inside one lineage
            lin = lineages[int(rng.integers(0, len(lineages)))]
            src = int(rng.integers(0, len(lin)))
            lin[src]["power2"] -= 1
            lin.insert(src + 1, _copy_graph(lin[src]))
            grow -= 1
        levels.append(emit(j + 1))
    return CompiledComponent(tuple(int(i) for i in output_indices), np.asarray(f_selection, np.int32), tuple(levels))


def physical_program(*, num_f: int, n_direct: int, components: list[dict], seed: int = 42,
                     num_detectors: int | None = None, shuffle_outputs: bool = False,
                     direct_flip_fraction: float = 0.0, identity_direct: bool = True,
                     approx: bool = False, live_padding: bool = False) -> CompiledProgram:
    """Normalised counterpart of :func:`synth_program` (same arguments, same shapes)."""
    rng = np.random.default_rng([seed, 0x70687973])
    n_comp_out = sum(c["n"] for c in components)
    num_outputs = n_direct + n_comp_out
    out_ids = rng.permutation(num_outputs) if shuffle_outputs else np.arange(num_outputs)
    dfi = np.arange(n_direct) if identity_direct else (rng.choice(num_f, size=n_direct, replace=False) if n_direct else np.zeros(0, int))
    flips = rng.random(n_direct) < direct_flip_fraction
    direct = [(int(out_ids[j]), int(dfi[j]), bool(flips[j])) for j in range(n_direct)]
    comps, pos = [], n_direct
    for c in components:
        n, F = int(c["n"]), int(c["F"])
        kw = {k: v for k, v in c.items() if k in ("density", "ta", "tb", "tc", "td", "shared_delta")}
        fsel = np.sort(rng.choice(num_f, size=F, replace=False)).astype(np.int32)
        comps.append(physical_component(rng, out_ids[pos:pos + n], fsel, list(c["G"]), approx=approx, live_padding=live_padding, **kw))
        pos += n
    nd = num_detectors if num_detectors is not None else n_direct
    return make_program(comps, direct, num_outputs, nd)


# --------------------------------------------------------------------------
# shape classes around the BASELINE estimates (scripts/shape_map.py, tests/test_gpu_shape_classes.py)
# --------------------------------------------------------------------------
#
# SURVEY 8(a) marks F, T_x, n_direct and num_f of every BASELINE circuit as [ESTIMATE]: real compiled programs may land
# anywhere around the five configurations above, and the reference treats every shape alike (sampler.py:117-167,
# compile/pipeline.py:55-102).  A class = one wall of the kernels' eligibility rules moved on its own from the nearest
# configuration (`near`): f-row width, output count, components (number, narrow / wide mix), outputs per component,
# selected bits per component.


def _graph_counts(n: int, top: int) -> list[int]:
    """n + 1 non-decreasing graph counts growing to ``top`` (the C2 profile, scaled)."""
    g = [max(1, int(round(top * (k + 1) / (n + 1)))) for k in range(n + 1)]
    for k in range(1, n + 1):
        g[k] = max(g[k], g[k - 1])
    return g


def _narrow(n: int, F: int, top: int = 40, **kw) -> dict:
    return dict(n=n, F=F, G=_graph_counts(n, top), **kw)


def _wide(n: int, F: int) -> dict:
    return dict(n=n, F=F, G=[min(5, 1 + (k + 1) // 2) for k in range(n + 1)], density=0.08)


SHAPE_CLASSES = {
    # ---- the f-row width wall (narrow component inside ever wider rows; C2 otherwise)
    "f64": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(5, 32)]),
    "f128": dict(near="C2", num_f=128, n_direct=15, components=[_narrow(5, 32)]),
    "f160": dict(near="C2", num_f=160, n_direct=15, components=[_narrow(5, 32)]),
    "f320": dict(near="C2", num_f=320, n_direct=15, components=[_narrow(5, 32)]),
    "f600": dict(near="C2", num_f=600, n_direct=15, components=[_narrow(5, 32)]),
    # ---- the output-count wall
    "out64": dict(near="C3", num_f=128, n_direct=59, components=[_narrow(5, 32)]),
    "out65": dict(near="C3", num_f=128, n_direct=60, components=[_narrow(5, 32)]),
    "out121": dict(near="C3", num_f=128, n_direct=116, components=[_narrow(5, 32)]),
    "out121_f320": dict(near="C5", num_f=320, n_direct=118, components=[_narrow(3, 30, 8)]),  # the plausible REAL d = 5 surface + T shape
    "out260_f320": dict(near="C5", num_f=320, n_direct=255, components=[_narrow(5, 32)]),
    # ---- components: number and mix
    "3narrow": dict(near="C4", num_f=64, n_direct=10, components=[_narrow(1, 12, 6), _narrow(1, 12, 6), _narrow(6, 40, 64)]),
    "6narrow": dict(near="C4", num_f=64, n_direct=10, components=[_narrow(1, 8, 4), _narrow(1, 8, 4), _narrow(1, 10, 4), _narrow(2, 12, 8), _narrow(2, 16, 8), _narrow(3, 24, 16)]),
    "6narrow_f320": dict(near="C4", num_f=320, n_direct=100, components=[_narrow(1, 8, 4), _narrow(1, 8, 4), _narrow(1, 10, 4), _narrow(2, 12, 8), _narrow(2, 16, 8), _narrow(3, 24, 16)]),
    "1wide": dict(near="C5", num_f=320, n_direct=118, components=[_wide(3, 200)]),
    "2wide": dict(near="C5", num_f=320, n_direct=118, components=[_wide(2, 100), _wide(3, 150)]),
    "narrow+wide": dict(near="C5", num_f=320, n_direct=118, components=[_narrow(2, 20, 8), _wide(3, 200)]),
    # ---- outputs per component
    "n1": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(1, 32, 8)]),
    "n8": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(8, 32)]),
    "n9": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(9, 32)]),
    "n11": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(11, 32)]),
    "n17total": dict(near="C4", num_f=64, n_direct=10, components=[_narrow(5, 20, 24), _narrow(6, 20, 24), _narrow(6, 20, 24)]),
    # ---- selected bits per component
    "F16": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(5, 16)]),
    "F59": dict(near="C2", num_f=64, n_direct=5, components=[_narrow(5, 59)]),
    "F60": dict(near="C2", num_f=64, n_direct=4, components=[_narrow(5, 60)]),  # F + n = 65 parameters: a "wide" component in narrow rows
    "F70": dict(near="C2", num_f=96, n_direct=4, components=[_narrow(5, 70)]),  # more than 64 SELECTED bits and 140 graphs: the wall after F60's (x in three words serves F <= 64)
    "F140": dict(near="C2", num_f=160, n_direct=4, components=[_narrow(5, 140)]),  # beyond 128 parameters with 140 graphs: the wall behind F70's (x in four words)
    "F255": dict(near="C5", num_f=320, n_direct=65, components=[_wide(3, 255)]),
    "F300": dict(near="C5", num_f=320, n_direct=20, components=[_wide(3, 300)]),
    "F200_f600": dict(near="C5", num_f=600, n_direct=118, components=[_wide(3, 200)]),  # max_f_index >= 512
    # ---- round 6: beyond the compiled-in walls - the reference loops over any number of levels
    # (sampler.py:62) and components are whatever compile/pipeline.py:55-102 finds connected
    "n13": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(13, 32)]),
    "n16": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(16, 32)]),
    # (n24 / n40: four of five outputs are deterministic parities of f and the earlier outcomes - see physical_component;
    #  n24x is the unconstrained mixture, whose int32 coefficients wrap like the reference's would: nonsense marginals, parity only)
    "n24": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(24, 32, shared_delta=0.8)]),
    "n40": dict(near="C2", num_f=64, n_direct=8, components=[_narrow(40, 32, shared_delta=0.8)]),
    "n24x": dict(near="C2", num_f=64, n_direct=15, components=[_narrow(24, 32)]),
    "w12": dict(near="C5", num_f=320, n_direct=100, components=[_wide(12, 200)]),  # a wide component of 12 outputs
    "20narrow": dict(near="C4", num_f=192, n_direct=10,
                     components=[_narrow(1 + (k % 3), 6 + (k % 5), 4 + 2 * (k % 4)) for k in range(20)]),  # more than 16 components
    "9wide": dict(near="C5", num_f=704, n_direct=60, components=[_wide(2, 70) for _ in range(9)]),  # more than 8 wide passes
    "F600": dict(near="C5", num_f=640, n_direct=20, components=[_wide(3, 600)]),  # >= 512 selected bits in one component
}


def shape_class_program(name: str, *, seed: int = 42) -> tuple[CompiledProgram, dict]:
    """``(program, class dict)`` of one entry of ``SHAPE_CLASSES`` (normalised probability model, shuffled output columns off)."""
    c = SHAPE_CLASSES[name]
    prog = physical_program(num_f=c["num_f"], n_direct=c["n_direct"], components=[dict(x) for x in c["components"]], seed=seed)
    return prog, dict(c, name=name, p_bit=0.02)
