"""A stand-in for a live ``tsim`` compiled program, for exercising the exporter / drop-in seam without tsim.

The classes mirror the reference's containers structurally (/root/reference/src/tsim/core/types.py:55-107,
compile/compile.py:21-37, compile/terms.py:42-207): frozen attribute objects, leaves that are NOT numpy
arrays (``ForeignArray`` only offers ``__array__``/``shape``/``dtype`` like a jax.Array), ``output_indices``
as a static tuple, ``has_approximate_floatfactors`` as a static bool, ``output_reindex`` possibly None.
"""

from __future__ import annotations

import sys
import types
from dataclasses import dataclass

import numpy as np


class ForeignArray:
    """Array-like that is not an ndarray (conversion must go through ``np.asarray``)."""

    def __init__(self, a, dtype):
        self._a = np.array(a, dtype=dtype)
        self._a.setflags(write=False)

    @property
    def shape(self):
        return self._a.shape

    @property
    def dtype(self):
        return self._a.dtype

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __len__(self):
        return len(self._a)

    def __getitem__(self, k):
        return self._a[k]


class ForeignScalar:
    """A 0-d array-like leaf (what ``jnp.asarray(5)`` or a traced static looks like from outside): converts through
    ``int()`` / ``bool()`` / ``np.asarray`` only."""

    def __init__(self, v):
        self._v = v

    def __int__(self):
        return int(self._v)

    def __index__(self):
        return int(self._v)

    def __bool__(self):
        return bool(self._v)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._v, dtype=dtype)


@dataclass(frozen=True)
class FNodePhases:
    phases: ForeignArray
    params: ForeignArray
    counts: ForeignArray


@dataclass(frozen=True)
class FHalfPiPhases:
    coeffs: ForeignArray
    params: ForeignArray


@dataclass(frozen=True)
class FPiProducts:
    psi_const: ForeignArray
    psi_params: ForeignArray
    phi_const: ForeignArray
    phi_params: ForeignArray


@dataclass(frozen=True)
class FPhasePairs:
    alpha: ForeignArray
    alpha_params: ForeignArray
    beta: ForeignArray
    beta_params: ForeignArray
    counts: ForeignArray


@dataclass(frozen=True)
class FScalarPrefactor:
    phase_indices: ForeignArray
    floatfactor: ForeignArray
    power2: ForeignArray
    approximate_floatfactors: ForeignArray
    has_approximate_floatfactors: bool


@dataclass(frozen=True)
class FCompiledScalarGraphs:
    num_graphs: int
    n_params: int
    node_phases: FNodePhases
    halfpi_phases: FHalfPiPhases
    pi_products: FPiProducts
    phase_pairs: FPhasePairs
    prefactor: FScalarPrefactor


@dataclass(frozen=True)
class FCompiledComponent:
    output_indices: tuple
    f_selection: ForeignArray
    compiled_scalar_graphs: tuple


@dataclass(frozen=True)
class FCompiledProgram:
    components: tuple
    direct_f_indices: ForeignArray
    direct_flips: ForeignArray
    output_order: ForeignArray
    output_reindex: object
    num_outputs: int
    num_detectors: int


def to_foreign(program, scalar_leaves: bool = False) -> FCompiledProgram:
    """Re-express one of this repo's plain programs in the foreign containers (dtypes as the reference's).
    ``scalar_leaves``: counts and flags as 0-d array-likes instead of Python ints / bools."""
    A = ForeignArray
    S = ForeignScalar if scalar_leaves else (lambda v: v)

    def level(lv):
        a, b, c, d, p = lv.node_phases, lv.halfpi_phases, lv.pi_products, lv.phase_pairs, lv.prefactor
        return FCompiledScalarGraphs(
            S(int(lv.num_graphs)), S(int(lv.n_params)),
            FNodePhases(A(a.phases, np.uint8), A(a.params, np.uint8), A(a.counts, np.int32)),
            FHalfPiPhases(A(b.coeffs, np.uint8), A(b.params, np.uint8)),
            FPiProducts(A(c.psi_const, np.uint8), A(c.psi_params, np.uint8), A(c.phi_const, np.uint8), A(c.phi_params, np.uint8)),
            FPhasePairs(A(d.alpha, np.uint8), A(d.alpha_params, np.uint8), A(d.beta, np.uint8), A(d.beta_params, np.uint8),
                        A(d.counts, np.int32)),
            FScalarPrefactor(A(p.phase_indices, np.uint8), A(p.floatfactor, np.int32), A(p.power2, np.int32),
                             A(p.approximate_floatfactors, np.complex64), S(bool(p.has_approximate_floatfactors))),
        )

    comps = tuple(
        FCompiledComponent(tuple(int(i) for i in c.output_indices), A(c.f_selection, np.int32),
                           tuple(level(lv) for lv in c.compiled_scalar_graphs))
        for c in program.components
    )
    return FCompiledProgram(
        comps, A(program.direct_f_indices, np.int32), A(program.direct_flips, np.bool_), A(program.output_order, np.int32),
        None if program.output_reindex is None else A(program.output_reindex, np.int32),
        S(int(program.num_outputs)), S(int(program.num_detectors)),
    )


class fake_tsim:
    """Context manager: a minimal ``tsim`` package in ``sys.modules`` whose ``tsim.sampler.sample_program`` is a
    module global resolved at call time, like the reference's (src/tsim/sampler.py:117,274,400,484)."""

    def __enter__(self):
        self._saved = {k: sys.modules.get(k) for k in ("tsim", "tsim.sampler")}
        pkg = types.ModuleType("tsim")
        pkg.__path__ = []
        smod = types.ModuleType("tsim.sampler")

        def sample_program(program, f_params, key):  # the "JAX" implementation a user would be replacing
            raise RuntimeError("reference sample_program called: the backend was not installed")

        def evaluate(circuit, param_vals):
            raise RuntimeError("reference evaluate called")

        def run(program, f_params, key):  # a call site: resolves the module global at call time
            return sys.modules["tsim.sampler"].sample_program(program, f_params, key)

        smod.sample_program, smod.evaluate, smod.run = sample_program, evaluate, run
        pkg.sampler = smod
        sys.modules["tsim"], sys.modules["tsim.sampler"] = pkg, smod
        return smod

    def __exit__(self, *exc):
        for k, v in self._saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
