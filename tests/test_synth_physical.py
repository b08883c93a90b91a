"""The normalised synthetic programs (tsim_amd.synth.physical_program) ARE probability models.

Checked with the oracle's exact integer arithmetic (no floats): for every level i and random contexts
    amp_i(f, m_<i, 0) + amp_i(f, m_<i, 1) == amp_{i-1}(f, m_<i)      in Z[w] * 2^k,
amplitudes are real and non-negative, and the sampler's normalisation deviation is float32 rounding.
"""

import warnings

import numpy as np
import pytest

from oracle import oracle_np as O
from tsim_amd import synth


def _exact(level, pv):
    """Exact value as python ints: (a, b, c, d, power) with odd-gcd coefficients."""
    coeffs, power = O.evaluate_exact(level, pv)
    out = []
    for c, p in zip(coeffs.tolist(), power.tolist()):
        if not any(c):
            out.append((0, 0, 0, 0, 0))
            continue
        while all(v % 2 == 0 for v in c):
            c = [v // 2 for v in c]
            p += 1
        out.append((*c, p))
    return out


def _add(x, y):
    if not any(x[:4]):
        return y
    if not any(y[:4]):
        return x
    p = min(x[4], y[4])
    c = [a * (1 << (x[4] - p)) + b * (1 << (y[4] - p)) for a, b in zip(x[:4], y[:4])]
    if not any(c):
        return (0, 0, 0, 0, 0)
    while all(v % 2 == 0 for v in c):
        c = [v // 2 for v in c]
        p += 1
    return (*c, p)


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5"])
def test_levels_are_exact_marginals(name):
    prog, cfg = synth.config_program(name)
    rng = np.random.default_rng(1)
    nrows = 40 if name == "C4" else 120
    for comp in prog.components:
        F, n = len(comp.f_selection), len(comp.output_indices)
        levels = comp.compiled_scalar_graphs
        f = (rng.random((nrows, F)) < 0.15).astype(np.uint8)
        m = rng.integers(0, 2, size=(nrows, n), dtype=np.uint8)
        prev = _exact(levels[0], f)
        for v in prev:  # real, positive: b == d (w + conj w) and c == 0 on the basis (1, w, i, conj w)
            assert v[1] == v[3] and v[2] == 0
        for i in range(n):
            ctx = np.concatenate([f, m[:, :i]], axis=1)
            a0 = _exact(levels[i + 1], np.concatenate([ctx, np.zeros((nrows, 1), np.uint8)], axis=1))
            a1 = _exact(levels[i + 1], np.concatenate([ctx, np.ones((nrows, 1), np.uint8)], axis=1))
            for r in range(nrows):
                assert _add(a0[r], a1[r]) == prev[r], (name, i, r)
            z0 = O.evaluate(levels[i + 1], np.concatenate([ctx, np.zeros((nrows, 1), np.uint8)], axis=1))
            z1 = O.evaluate(levels[i + 1], np.concatenate([ctx, np.ones((nrows, 1), np.uint8)], axis=1))
            assert (z0.real >= 0).all() and (z1.real >= 0).all()
            assert np.abs(z0.imag).max() <= 1e-6 * max(1.0, np.abs(z0.real).max())
            prev = [a1[r] if m[r, i] else a0[r] for r in range(nrows)]


@pytest.mark.parametrize("name", ["C2", "C3"])
def test_live_padding_keeps_the_model_and_survives_the_algebra(name):
    """live_padding: the exact marginal identity still holds (the common phase w^q(f) multiplies every level alike),
    amplitudes are genuinely complex, the normalisation deviation stays at float32 rounding."""
    prog, cfg = synth.config_program(name, live_padding=True)
    rng = np.random.default_rng(5)
    comp = prog.components[0]
    F, n = len(comp.f_selection), len(comp.output_indices)
    levels = comp.compiled_scalar_graphs
    nrows = 60
    f = (rng.random((nrows, F)) < 0.15).astype(np.uint8)
    m = rng.integers(0, 2, size=(nrows, n), dtype=np.uint8)
    prev = _exact(levels[0], f)
    assert any(v[1] != v[3] or v[2] != 0 for v in prev)  # not all real any more
    for i in range(n):
        ctx = np.concatenate([f, m[:, :i]], axis=1)
        a0 = _exact(levels[i + 1], np.concatenate([ctx, np.zeros((nrows, 1), np.uint8)], axis=1))
        a1 = _exact(levels[i + 1], np.concatenate([ctx, np.ones((nrows, 1), np.uint8)], axis=1))
        for r in range(nrows):
            assert _add(a0[r], a1[r]) == prev[r], (name, i, r)
        prev = [a1[r] if m[r, i] else a0[r] for r in range(nrows)]
    fb = synth.synth_f(300, cfg["num_f"], cfg["p_bit"], seed=3)
    _, devs = O.sample_program(prog, fb, (3, 4), return_devs=True)
    assert max(float(d) for d in devs) < 1e-5
    # same graph counts as the neutral-padding program, comparable term counts
    ref, _ = synth.config_program(name)
    assert [lv.num_graphs for lv in levels] == [lv.num_graphs for lv in ref.components[0].compiled_scalar_graphs]
    assert (levels[-1].halfpi_phases.coeffs != 0).sum() > 0 and levels[-1].pi_products.psi_params.any()


@pytest.mark.parametrize("name,approx", [("C2", False), ("C2", True), ("C3", False), ("C5", False)])
def test_no_normalisation_warning(name, approx):
    from tsim_amd.backend import check_norm_deviation

    prog, cfg = synth.config_program(name, approx=approx)
    f = synth.synth_f(400, cfg["num_f"], cfg["p_bit"], seed=3)
    _, devs = O.sample_program(prog, f, (3, 4), return_devs=True)
    assert max(float(d) for d in devs) < 1e-5
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for d in devs:
            check_norm_deviation(float(d))


def test_shapes_match_the_published_ones():
    for name in ("C2", "C3", "C4", "C5"):
        a, _ = synth.config_program(name, physical=True)
        b, _ = synth.config_program(name, physical=False)
        for ca, cb in zip(a.components, b.components):
            assert [lv.num_graphs for lv in ca.compiled_scalar_graphs] == [lv.num_graphs for lv in cb.compiled_scalar_graphs]
            assert [lv.n_params for lv in ca.compiled_scalar_graphs] == [lv.n_params for lv in cb.compiled_scalar_graphs]
            assert len(ca.f_selection) == len(cb.f_selection)
        assert a.num_outputs == b.num_outputs and len(a.direct_f_indices) == len(b.direct_f_indices)
    # every term family is present in the default benchmark program
    lv = synth.config_program("C2")[0].components[0].compiled_scalar_graphs[-1]
    assert lv.node_phases.counts.max() > 0 and (lv.halfpi_phases.coeffs != 0).any()
    assert lv.pi_products.psi_params.any() and lv.phase_pairs.counts.max() > 0
