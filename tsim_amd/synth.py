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


def config_program(name: str, *, approx: bool = False) -> tuple[CompiledProgram, dict]:
    """Return ``(program, config_dict)`` for one of ``C1..C5``."""
    cfg = CONFIGS[name]
    comps = [dict(c, approx=approx) if approx else dict(c) for c in cfg["components"]]
    prog = synth_program(
        num_f=cfg["num_f"], n_direct=cfg["n_direct"], components=comps, seed=cfg["seed"]
    )
    return prog, cfg
