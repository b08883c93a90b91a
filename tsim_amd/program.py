"""Plain-data compiled-program container (the drop-in boundary types).

Mirrors, field for field and with the same names, the reference's compiled
data model so that a real ``tsim`` ``CompiledProgram`` can be converted with
``from_tsim`` (``np.asarray`` of every leaf) and so that the sampler code reads
like the reference's:

* ``CompiledProgram`` / ``CompiledComponent``  -> /root/reference/src/tsim/core/types.py:55-107
* ``CompiledScalarGraphs``                     -> /root/reference/src/tsim/compile/compile.py:21-37
* ``NodePhases`` / ``HalfPiPhases`` / ``PiProducts`` / ``PhasePairs`` /
  ``ScalarPrefactor``                          -> /root/reference/src/tsim/compile/terms.py:42-207

All arrays are numpy, one *byte per bit*, row-major, padded to the per-family
maximum term count - exactly the reference layout.  Bit-packing happens behind
the C-ABI (``tsim_program_add_level``), never here.

No arithmetic lives in this module: it is containers, validation and ``.npz``
(de)serialisation only.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np


def _u8(a, shape_tail=None) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.uint8)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


@dataclass
class NodePhases:
    """``Π_t (1 + ω^(4·parity_t + phase_t))`` (terms.py:42-73)."""

    phases: np.ndarray  # uint8 [G, T_A], 0..7
    params: np.ndarray  # uint8 [G, T_A, P]
    counts: np.ndarray  # int32 [G]


@dataclass
class HalfPiPhases:
    """``ω^(Σ_t coeff_t·parity_t)`` (terms.py:76-107)."""

    coeffs: np.ndarray  # uint8 [G, T_B], {0,2,4,6}
    params: np.ndarray  # uint8 [G, T_B, P]


@dataclass
class PiProducts:
    """``(-1)^(Σ_t ψ_t·φ_t)`` (terms.py:110-144)."""

    psi_const: np.ndarray  # uint8 [G, T_C]
    psi_params: np.ndarray  # uint8 [G, T_C, P]
    phi_const: np.ndarray  # uint8 [G, T_C]
    phi_params: np.ndarray  # uint8 [G, T_C, P]


@dataclass
class PhasePairs:
    """``Π_t (1 + ω^α + ω^β − ω^(α+β))`` (terms.py:147-187)."""

    alpha: np.ndarray  # uint8 [G, T_D]
    alpha_params: np.ndarray  # uint8 [G, T_D, P]
    beta: np.ndarray  # uint8 [G, T_D]
    beta_params: np.ndarray  # uint8 [G, T_D, P]
    counts: np.ndarray  # int32 [G]


@dataclass
class ScalarPrefactor:
    """Per-graph static prefactor (terms.py:190-207)."""

    phase_indices: np.ndarray  # uint8 [G]
    floatfactor: np.ndarray  # int32 [G, 4]
    power2: np.ndarray  # int32 [G]
    approximate_floatfactors: np.ndarray  # complex64 [G]
    has_approximate_floatfactors: bool = False


@dataclass
class CompiledScalarGraphs:
    """One autoregressive level: a list of ``num_graphs`` scalar terms."""

    num_graphs: int
    n_params: int
    node_phases: NodePhases
    halfpi_phases: HalfPiPhases
    pi_products: PiProducts
    phase_pairs: PhasePairs
    prefactor: ScalarPrefactor


@dataclass
class CompiledComponent:
    """A connected component (types.py:55-77)."""

    output_indices: tuple
    f_selection: np.ndarray  # int32 [F]
    compiled_scalar_graphs: tuple


@dataclass
class CompiledProgram:
    """The full program (types.py:80-107)."""

    components: tuple
    direct_f_indices: np.ndarray  # int32 [n_direct]
    direct_flips: np.ndarray  # bool [n_direct]
    output_order: np.ndarray  # int32 [num_outputs]
    output_reindex: np.ndarray | None
    num_outputs: int
    num_detectors: int
    # cache slot used by the HIP backend for the uploaded device handle
    _backend_cache: dict = field(default_factory=dict, repr=False, compare=False)


# --------------------------------------------------------------------------
# construction helpers
# --------------------------------------------------------------------------


def empty_scalar_graphs(n_params: int) -> CompiledScalarGraphs:
    """The ``compile_scalar_graphs([], params)`` result (test_compile.py:13-46)."""
    P = n_params
    return CompiledScalarGraphs(
        num_graphs=0,
        n_params=P,
        node_phases=NodePhases(
            np.zeros((0, 0), np.uint8), np.zeros((0, 0, P), np.uint8), np.zeros((0,), np.int32)
        ),
        halfpi_phases=HalfPiPhases(np.zeros((0, 0), np.uint8), np.zeros((0, 0, P), np.uint8)),
        pi_products=PiProducts(
            np.zeros((0, 0), np.uint8),
            np.zeros((0, 0, P), np.uint8),
            np.zeros((0, 0), np.uint8),
            np.zeros((0, 0, P), np.uint8),
        ),
        phase_pairs=PhasePairs(
            np.zeros((0, 0), np.uint8),
            np.zeros((0, 0, P), np.uint8),
            np.zeros((0, 0), np.uint8),
            np.zeros((0, 0, P), np.uint8),
            np.zeros((0,), np.int32),
        ),
        prefactor=ScalarPrefactor(
            np.zeros((0,), np.uint8),
            np.zeros((0, 4), np.int32),
            np.zeros((0,), np.int32),
            np.zeros((0,), np.complex64),
            False,
        ),
    )


def scalar_graphs_from_terms(n_params: int, graphs: list[dict]) -> CompiledScalarGraphs:
    """Build a padded ``CompiledScalarGraphs`` from per-graph term lists.

    ``graphs[g]`` is a dict with optional keys

    * ``"A"``: list of ``(phase, bits)``                      (NodePhases)
    * ``"B"``: list of ``(coeff, bits)``                      (HalfPiPhases)
    * ``"C"``: list of ``(psi_const, psi_bits, phi_const, phi_bits)``
    * ``"D"``: list of ``(alpha, alpha_bits, beta, beta_bits)``
    * ``"phase"``: int 0..7, ``"floatfactor"``: 4 ints, ``"power2"``: int,
      ``"approx"``: complex

    where ``bits`` is an iterable of parameter indices whose XOR forms the
    parity.  Padding follows compile.py:40-238 (zeros; counts for A and D).
    """
    G = len(graphs)
    P = n_params
    TA = max([len(g.get("A", [])) for g in graphs], default=0)
    TB = max([len(g.get("B", [])) for g in graphs], default=0)
    TC = max([len(g.get("C", [])) for g in graphs], default=0)
    TD = max([len(g.get("D", [])) for g in graphs], default=0)
    a = NodePhases(np.zeros((G, TA), np.uint8), np.zeros((G, TA, P), np.uint8), np.zeros(G, np.int32))
    b = HalfPiPhases(np.zeros((G, TB), np.uint8), np.zeros((G, TB, P), np.uint8))
    c = PiProducts(
        np.zeros((G, TC), np.uint8),
        np.zeros((G, TC, P), np.uint8),
        np.zeros((G, TC), np.uint8),
        np.zeros((G, TC, P), np.uint8),
    )
    d = PhasePairs(
        np.zeros((G, TD), np.uint8),
        np.zeros((G, TD, P), np.uint8),
        np.zeros((G, TD), np.uint8),
        np.zeros((G, TD, P), np.uint8),
        np.zeros(G, np.int32),
    )
    pre = ScalarPrefactor(
        np.zeros(G, np.uint8),
        np.zeros((G, 4), np.int32),
        np.zeros(G, np.int32),
        np.ones(G, np.complex64),
        False,
    )

    def setbits(row, bits):
        for i in bits:
            if not 0 <= int(i) < P:
                raise ValueError(f"parameter index {i} out of range for n_params={P}")
            row[int(i)] ^= 1

    for gi, g in enumerate(graphs):
        for t, (ph, bits) in enumerate(g.get("A", [])):
            a.phases[gi, t] = ph % 8
            setbits(a.params[gi, t], bits)
        a.counts[gi] = len(g.get("A", []))
        for t, (co, bits) in enumerate(g.get("B", [])):
            b.coeffs[gi, t] = co % 8
            setbits(b.params[gi, t], bits)
        for t, (pc, pb, qc, qb) in enumerate(g.get("C", [])):
            c.psi_const[gi, t] = pc & 1
            setbits(c.psi_params[gi, t], pb)
            c.phi_const[gi, t] = qc & 1
            setbits(c.phi_params[gi, t], qb)
        for t, (al, ab, be, bb) in enumerate(g.get("D", [])):
            d.alpha[gi, t] = al % 8
            setbits(d.alpha_params[gi, t], ab)
            d.beta[gi, t] = be % 8
            setbits(d.beta_params[gi, t], bb)
        d.counts[gi] = len(g.get("D", []))
        pre.phase_indices[gi] = g.get("phase", 0) % 8
        pre.floatfactor[gi] = g.get("floatfactor", (1, 0, 0, 0))
        pre.power2[gi] = g.get("power2", 0)
        pre.approximate_floatfactors[gi] = g.get("approx", 1.0)
    pre.has_approximate_floatfactors = bool(np.any(pre.approximate_floatfactors != 1.0))
    return CompiledScalarGraphs(G, P, a, b, c, d, pre)


def make_program(
    components: list[CompiledComponent],
    direct: list[tuple[int, int, bool]],
    num_outputs: int,
    num_detectors: int,
) -> CompiledProgram:
    """Assemble a program the way ``compile_program`` does (pipeline.py:65-102).

    ``direct`` is a list of ``(output_idx, f_idx, flip)``; it is sorted by output
    index; components are ordered by their number of outputs (stable).
    """
    comps = sorted(components, key=lambda c: len(c.output_indices))
    direct = sorted(direct)
    order = [e[0] for e in direct]
    for c in comps:
        order.extend(c.output_indices)
    order = np.asarray(order, dtype=np.int32)
    if len(order) != num_outputs or sorted(order.tolist()) != list(range(num_outputs)):
        raise ValueError("outputs must be covered exactly once by direct entries and components")
    reindex = np.argsort(order).astype(np.int32)
    ident = np.array_equal(reindex, np.arange(num_outputs))
    return CompiledProgram(
        components=tuple(comps),
        direct_f_indices=np.asarray([e[1] for e in direct], dtype=np.int32),
        direct_flips=np.asarray([e[2] for e in direct], dtype=np.bool_),
        output_order=order,
        output_reindex=None if ident else reindex,
        num_outputs=int(num_outputs),
        num_detectors=int(num_detectors),
    )


# --------------------------------------------------------------------------
# validation
# --------------------------------------------------------------------------


def validate_scalar_graphs(c: CompiledScalarGraphs) -> None:
    G, P = c.num_graphs, c.n_params

    def chk(name, arr, shape, dtype=None):
        arr = np.asarray(arr)
        if arr.shape != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {arr.shape}")

    TA = np.asarray(c.node_phases.phases).shape[1] if G else 0
    TB = np.asarray(c.halfpi_phases.coeffs).shape[1] if G else 0
    TC = np.asarray(c.pi_products.psi_const).shape[1] if G else 0
    TD = np.asarray(c.phase_pairs.alpha).shape[1] if G else 0
    if G == 0:
        return
    chk("node_phases.phases", c.node_phases.phases, (G, TA))
    chk("node_phases.params", c.node_phases.params, (G, TA, P))
    chk("node_phases.counts", c.node_phases.counts, (G,))
    chk("halfpi_phases.coeffs", c.halfpi_phases.coeffs, (G, TB))
    chk("halfpi_phases.params", c.halfpi_phases.params, (G, TB, P))
    chk("pi_products.psi_const", c.pi_products.psi_const, (G, TC))
    chk("pi_products.psi_params", c.pi_products.psi_params, (G, TC, P))
    chk("pi_products.phi_const", c.pi_products.phi_const, (G, TC))
    chk("pi_products.phi_params", c.pi_products.phi_params, (G, TC, P))
    chk("phase_pairs.alpha", c.phase_pairs.alpha, (G, TD))
    chk("phase_pairs.alpha_params", c.phase_pairs.alpha_params, (G, TD, P))
    chk("phase_pairs.beta", c.phase_pairs.beta, (G, TD))
    chk("phase_pairs.beta_params", c.phase_pairs.beta_params, (G, TD, P))
    chk("phase_pairs.counts", c.phase_pairs.counts, (G,))
    chk("prefactor.phase_indices", c.prefactor.phase_indices, (G,))
    chk("prefactor.floatfactor", c.prefactor.floatfactor, (G, 4))
    chk("prefactor.power2", c.prefactor.power2, (G,))
    chk("prefactor.approximate_floatfactors", c.prefactor.approximate_floatfactors, (G,))


def validate_program(p: CompiledProgram, num_f: int | None = None) -> None:
    """Shape/range validation of a program (raises ``ValueError``)."""
    seen: list[int] = []
    nd = len(np.asarray(p.direct_f_indices))
    if len(np.asarray(p.direct_flips)) != nd:
        raise ValueError("direct_flips and direct_f_indices differ in length")
    if num_f is not None and nd and int(np.max(p.direct_f_indices)) >= num_f:
        raise ValueError("direct_f_indices out of range")
    for ci, comp in enumerate(p.components):
        n = len(comp.output_indices)
        levels = comp.compiled_scalar_graphs
        F = len(np.asarray(comp.f_selection))
        if num_f is not None and F and int(np.max(comp.f_selection)) >= num_f:
            raise ValueError(f"component {ci}: f_selection out of range")
        if len(levels) not in (n + 1, 2):
            raise ValueError(f"component {ci}: {len(levels)} levels for {n} outputs")
        sequential = len(levels) == n + 1
        for k, lv in enumerate(levels):
            want = F + (k if sequential else (0 if k == 0 else n))
            if lv.n_params != want:
                raise ValueError(
                    f"component {ci} level {k}: n_params={lv.n_params}, expected {want}"
                )
            validate_scalar_graphs(lv)
        seen.extend(comp.output_indices)
    order = np.asarray(p.output_order)
    if len(order) != p.num_outputs:
        raise ValueError("output_order length != num_outputs")
    if list(order[nd:]) != list(seen):
        raise ValueError("output_order tail must list component outputs in processing order")


# --------------------------------------------------------------------------
# conversion from a live tsim object (duck-typed; jax arrays -> numpy)
# --------------------------------------------------------------------------


def _np(x, dtype):
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def scalar_graphs_from_tsim(c: Any) -> CompiledScalarGraphs:
    """Convert a ``tsim.compile.compile.CompiledScalarGraphs`` (or our own)."""
    a, b, cc, d, pre = c.node_phases, c.halfpi_phases, c.pi_products, c.phase_pairs, c.prefactor
    return CompiledScalarGraphs(
        num_graphs=int(c.num_graphs),
        n_params=int(c.n_params),
        node_phases=NodePhases(_np(a.phases, np.uint8), _np(a.params, np.uint8), _np(a.counts, np.int32)),
        halfpi_phases=HalfPiPhases(_np(b.coeffs, np.uint8), _np(b.params, np.uint8)),
        pi_products=PiProducts(
            _np(cc.psi_const, np.uint8),
            _np(cc.psi_params, np.uint8),
            _np(cc.phi_const, np.uint8),
            _np(cc.phi_params, np.uint8),
        ),
        phase_pairs=PhasePairs(
            _np(d.alpha, np.uint8),
            _np(d.alpha_params, np.uint8),
            _np(d.beta, np.uint8),
            _np(d.beta_params, np.uint8),
            _np(d.counts, np.int32),
        ),
        prefactor=ScalarPrefactor(
            _np(pre.phase_indices, np.uint8),
            _np(pre.floatfactor, np.int32).reshape(-1, 4),
            _np(pre.power2, np.int32),
            _np(pre.approximate_floatfactors, np.complex64),
            bool(pre.has_approximate_floatfactors),
        ),
    )


def from_tsim(program: Any) -> CompiledProgram:
    """Convert a live ``tsim.core.types.CompiledProgram`` into plain numpy data."""
    if isinstance(program, CompiledProgram):
        return program
    comps = []
    for comp in program.components:
        comps.append(
            CompiledComponent(
                output_indices=tuple(int(i) for i in comp.output_indices),
                f_selection=_np(comp.f_selection, np.int32),
                compiled_scalar_graphs=tuple(
                    scalar_graphs_from_tsim(g) for g in comp.compiled_scalar_graphs
                ),
            )
        )
    reindex = program.output_reindex
    return CompiledProgram(
        components=tuple(comps),
        direct_f_indices=_np(program.direct_f_indices, np.int32),
        direct_flips=_np(program.direct_flips, np.bool_),
        output_order=_np(program.output_order, np.int32),
        output_reindex=None if reindex is None else _np(reindex, np.int32),
        num_outputs=int(program.num_outputs),
        num_detectors=int(program.num_detectors),
    )


# --------------------------------------------------------------------------
# .npz (de)serialisation - the exporter/importer of SURVEY §8(f) row 2
# --------------------------------------------------------------------------

_LEVEL_FIELDS = (
    ("a_phases", "node_phases", "phases"),
    ("a_params", "node_phases", "params"),
    ("a_counts", "node_phases", "counts"),
    ("b_coeffs", "halfpi_phases", "coeffs"),
    ("b_params", "halfpi_phases", "params"),
    ("c_psi_const", "pi_products", "psi_const"),
    ("c_psi_params", "pi_products", "psi_params"),
    ("c_phi_const", "pi_products", "phi_const"),
    ("c_phi_params", "pi_products", "phi_params"),
    ("d_alpha", "phase_pairs", "alpha"),
    ("d_alpha_params", "phase_pairs", "alpha_params"),
    ("d_beta", "phase_pairs", "beta"),
    ("d_beta_params", "phase_pairs", "beta_params"),
    ("d_counts", "phase_pairs", "counts"),
    ("p_phase_indices", "prefactor", "phase_indices"),
    ("p_floatfactor", "prefactor", "floatfactor"),
    ("p_power2", "prefactor", "power2"),
    ("p_approx", "prefactor", "approximate_floatfactors"),
)


def save_npz(path, program: CompiledProgram, **extra: np.ndarray) -> None:
    """Write a program (plus optional extra arrays, e.g. channel tables) to ``.npz``."""
    program = from_tsim(program)
    out: dict[str, np.ndarray] = {
        "num_outputs": np.int64(program.num_outputs),
        "num_detectors": np.int64(program.num_detectors),
        "direct_f_indices": program.direct_f_indices,
        "direct_flips": program.direct_flips,
        "output_order": program.output_order,
        "num_components": np.int64(len(program.components)),
    }
    for ci, comp in enumerate(program.components):
        out[f"c{ci}_output_indices"] = np.asarray(comp.output_indices, dtype=np.int32)
        out[f"c{ci}_f_selection"] = comp.f_selection
        out[f"c{ci}_num_levels"] = np.int64(len(comp.compiled_scalar_graphs))
        for k, lv in enumerate(comp.compiled_scalar_graphs):
            pre = f"c{ci}_l{k}_"
            out[pre + "num_graphs"] = np.int64(lv.num_graphs)
            out[pre + "n_params"] = np.int64(lv.n_params)
            out[pre + "has_approx"] = np.bool_(lv.prefactor.has_approximate_floatfactors)
            for key, fam, attr in _LEVEL_FIELDS:
                out[pre + key] = np.asarray(getattr(getattr(lv, fam), attr))
    for k, v in extra.items():
        out["x_" + k] = np.asarray(v)
    np.savez_compressed(path, **out)


def load_npz(path) -> tuple[CompiledProgram, dict[str, np.ndarray]]:
    """Inverse of :func:`save_npz`; returns ``(program, extra_arrays)``."""
    z = np.load(path, allow_pickle=False)
    comps = []
    for ci in range(int(z["num_components"])):
        levels = []
        for k in range(int(z[f"c{ci}_num_levels"])):
            pre = f"c{ci}_l{k}_"
            f = {key: z[pre + key] for key, _, _ in _LEVEL_FIELDS}
            levels.append(
                CompiledScalarGraphs(
                    num_graphs=int(z[pre + "num_graphs"]),
                    n_params=int(z[pre + "n_params"]),
                    node_phases=NodePhases(f["a_phases"], f["a_params"], f["a_counts"]),
                    halfpi_phases=HalfPiPhases(f["b_coeffs"], f["b_params"]),
                    pi_products=PiProducts(
                        f["c_psi_const"], f["c_psi_params"], f["c_phi_const"], f["c_phi_params"]
                    ),
                    phase_pairs=PhasePairs(
                        f["d_alpha"], f["d_alpha_params"], f["d_beta"], f["d_beta_params"], f["d_counts"]
                    ),
                    prefactor=ScalarPrefactor(
                        f["p_phase_indices"],
                        f["p_floatfactor"].reshape(-1, 4),
                        f["p_power2"],
                        f["p_approx"],
                        bool(z[pre + "has_approx"]),
                    ),
                )
            )
        comps.append(
            CompiledComponent(
                output_indices=tuple(int(i) for i in z[f"c{ci}_output_indices"]),
                f_selection=z[f"c{ci}_f_selection"].astype(np.int32),
                compiled_scalar_graphs=tuple(levels),
            )
        )
    order = z["output_order"].astype(np.int32)
    reindex = np.argsort(order).astype(np.int32)
    ident = np.array_equal(reindex, np.arange(len(order)))
    prog = CompiledProgram(
        components=tuple(comps),
        direct_f_indices=z["direct_f_indices"].astype(np.int32),
        direct_flips=z["direct_flips"].astype(np.bool_),
        output_order=order,
        output_reindex=None if ident else reindex,
        num_outputs=int(z["num_outputs"]),
        num_detectors=int(z["num_detectors"]),
    )
    extra = {k[2:]: z[k] for k in z.files if k.startswith("x_")}
    return prog, extra
