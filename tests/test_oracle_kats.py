"""Pin the oracle against every golden vector the reference's own tests hold for the hot path.

Each test names the reference test it restates (paths under /root/reference/test).  None of
them needs JAX: the hand-built programs in tsim_amd/synth.py carry the amplitudes of the KAT
circuits, and the closed-form family references are recomputed with numpy from the same seeds.
"""

import numpy as np
import pytest

from conftest import run_batches
from oracle import oracle_np as O
from tsim_amd import prng, synth
from tsim_amd.program import NodePhases, HalfPiPhases, PiProducts, PhasePairs, empty_scalar_graphs


def np_sample(program, f, key):
    return O.sample_program(program, f, key)


# ---- seeded sampler KATs -----------------------------------------------------


def test_seed_counts_48_53_52_50():
    """unit/test_sampler.py:223-233 (two fresh samplers give the same sequence)."""
    for _ in range(2):
        outs = run_batches(np_sample, synth.kat_h_m(), 0, [100] * 4)
        assert [int(o.sum()) for o in outs] == [48, 53, 52, 50]


def test_t_gate_9_of_100():
    """integration/test_sampler_circuits.py:40-49."""
    (o,) = run_batches(np_sample, synth.kat_t_gate(), 0, [100])
    assert int(o.sum()) == 9


def test_s_gate_48_of_100():
    """integration/test_sampler_circuits.py:52-61 (P = 1/2, same stream as H;M)."""
    (o,) = run_batches(np_sample, synth.kat_h_m(), 0, [100])
    assert int(o.sum()) == 48


def test_r_gate_7_4_0():
    """integration/test_sampler_circuits.py:90-105: key threading across three components."""
    (o,) = run_batches(np_sample, synth.kat_r_gate(), 0, [10])
    assert o.sum(axis=0).tolist() == [7, 4, 0]


def test_bell_48_and_correlated():
    """integration/test_sampler_circuits.py:10-22."""
    (o,) = run_batches(np_sample, synth.kat_bell(), 0, [100])
    assert np.array_equal(o[:, 0], o[:, 1]) and int(o[:, 0].sum()) == 48


def test_t_dag_and_s_dag_all_zero():
    """integration/test_sampler_circuits.py:64-87: P(1) = 0 -> no ones in 10 shots."""
    from tsim_amd.program import make_program

    prog = make_program([synth.single_output_component(0, zero=True)], [], 1, 0)
    (o,) = run_batches(np_sample, prog, 0, [10])
    assert int(o.sum()) == 0


def test_host_prng_matches_oracle_split():
    k = (0, 0)
    for _ in range(6):
        assert prng.split(k) == O.split(k)
        k = prng.split(k)[0]
    assert prng.key(5) == O.key(5) == (0, 5)


# ---- exact scalar -------------------------------------------------------------


def test_sum_reduces_while_adding():
    """unit/core/test_exact_scalar.py:67-84 (exact KAT)."""
    coeffs = np.array(
        [
            [[1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]],
            [[1, 0, 0, 0], [1, 0, 0, 0], [0, 2, 0, 0], [0, 2, 0, 0]],
        ]
    )
    powers = np.array([[0, 0, 0, 0], [3, 3, 2, 2]])
    s = O.ExactScalarArray(coeffs, powers).sum()
    assert np.array_equal(s.coeffs, [[1, 0, 0, 0], [1, 1, 0, 0]])
    assert np.array_equal(s.power, [2, 4])


def _rand_scalars(seed=0, n=100):
    return np.random.default_rng(seed).integers(-2, 2, size=(n, 4)).astype(np.int32)


def test_scalar_multiplication_matches_complex():
    """unit/core/test_exact_scalar.py:14-24."""
    s = _rand_scalars()
    d1, d2 = O.ExactScalarArray(s[0]), O.ExactScalarArray(s[1])
    assert np.allclose((d1 * d2).to_complex(), d1.to_complex() * d2.to_complex())


def test_prod_and_sum_match_complex():
    """unit/core/test_exact_scalar.py:27-64."""
    sc = _rand_scalars().reshape(10, 10, 4)
    arr = O.ExactScalarArray(sc)
    assert np.allclose(arr.prod(axis=1).to_complex(), np.prod(arr.to_complex(), axis=1), atol=1e-4)
    powers = np.tile(np.arange(10, dtype=np.int32), (10, 1))
    arr = O.ExactScalarArray(sc, powers)
    assert np.allclose(arr.sum().to_complex(), np.sum(arr.to_complex(), axis=-1), atol=1e-3)
    single = O.ExactScalarArray(np.array([[[1, 2, 0, -1]]])).prod(axis=1)
    assert np.array_equal(single.coeffs, [[1, 2, 0, -1]])


# ---- GF(2) contraction ----------------------------------------------------------


def test_matmul_gf2_matches_int64():
    """unit/utils/test_linalg.py:102-112."""
    np.random.seed(0)
    G, T, P, B = 3, 4, 17, 5
    a = np.random.randint(0, 2, size=(G, T, P), dtype=np.uint8)
    b = np.random.randint(0, 2, size=(B, P), dtype=np.uint8)
    want = ((b.astype(np.int64) @ a.astype(np.int64).reshape(G * T, P).T) % 2).reshape(B, G, T)
    assert np.array_equal(O.matmul_gf2(a, b), want.astype(np.uint8))


@pytest.mark.parametrize("p", [256, 300, 1024])
def test_matmul_gf2_no_saturation(p):
    """unit/utils/test_linalg.py:115-126."""
    assert O.matmul_gf2(np.ones((1, 1, p), np.uint8), np.ones((1, p), np.uint8)).flatten()[0] == p % 2


def test_matmul_gf2_empty():
    """unit/utils/test_linalg.py:129-133."""
    assert O.matmul_gf2(np.zeros((0, 0, 4), np.uint8), np.zeros((2, 4), np.uint8)).shape == (2, 0, 0)


# ---- term families against the closed forms of unit/compile/test_terms.py:8-48 ----


def _ref_parity(bits, pv):
    return ((pv @ bits.reshape(-1, bits.shape[-1]).T) % 2).reshape(pv.shape[0], bits.shape[0], bits.shape[1])


@pytest.mark.parametrize("seed", (0, 42))
def test_node_phases_closed_form(seed):
    np.random.seed(seed)
    G, T, P, B = 3, 4, 5, 7
    phases = np.random.randint(0, 8, size=(G, T)).astype(np.uint8)
    params = np.random.randint(0, 2, size=(G, T, P)).astype(np.uint8)
    counts = np.array([T, T - 1, 0], dtype=np.int32)
    pv = np.random.randint(0, 2, size=(B, P)).astype(np.uint8)
    got = O.node_phases_evaluate(NodePhases(phases, params, counts), pv).to_complex()
    term = 1 + np.exp(1j * np.pi * phases[None] / 4 + 1j * np.pi * _ref_parity(params, pv))
    mask = np.arange(T)[None, :] < counts[:, None]
    want = np.prod(np.where(mask[None], term, 1.0), axis=-1)
    np.testing.assert_allclose(got, want, atol=1e-5)


def test_node_phases_padding_and_empty():
    """test_terms.py:72-105: nonzero padded slots are masked; max_terms == 0 -> ones."""
    np.random.seed(0)
    G, P, B = 2, 4, 3
    counts = np.array([1, 2], dtype=np.int32)
    phases = np.concatenate([np.random.randint(0, 8, size=(G, 2)), np.full((G, 1), 5)], axis=1).astype(np.uint8)
    params = np.concatenate([np.random.randint(0, 2, size=(G, 2, P)), np.ones((G, 1, P))], axis=1).astype(np.uint8)
    pv = np.random.randint(0, 2, size=(B, P)).astype(np.uint8)
    got = O.node_phases_evaluate(NodePhases(phases, params, counts), pv).to_complex()
    term = 1 + np.exp(1j * np.pi * phases[None] / 4 + 1j * np.pi * _ref_parity(params, pv))
    mask = np.arange(3)[None, :] < counts[:, None]
    np.testing.assert_allclose(got, np.prod(np.where(mask[None], term, 1.0), axis=-1), atol=1e-5)
    e = O.node_phases_evaluate(
        NodePhases(np.zeros((2, 0), np.uint8), np.zeros((2, 0, 3), np.uint8), np.zeros(2, np.int32)),
        np.zeros((4, 3), np.uint8),
    ).to_complex()
    np.testing.assert_allclose(e, np.ones((4, 2)))


@pytest.mark.parametrize("seed", (0, 42))
def test_halfpi_closed_form(seed):
    np.random.seed(seed)
    G, T, P, B = 3, 4, 5, 6
    coeffs = np.random.choice([0, 2, 4, 6], size=(G, T)).astype(np.uint8)
    params = np.random.randint(0, 2, size=(G, T, P)).astype(np.uint8)
    pv = np.random.randint(0, 2, size=(B, P)).astype(np.uint8)
    got = O.halfpi_phases_evaluate(HalfPiPhases(coeffs, params), pv).to_complex()
    want = np.prod(np.exp(1j * np.pi * coeffs[None] * _ref_parity(params, pv) / 4), axis=-1)
    np.testing.assert_allclose(got, want, atol=1e-6)


@pytest.mark.parametrize("seed", (0, 42))
def test_pi_products_closed_form(seed):
    np.random.seed(seed)
    G, T, P, B = 3, 4, 5, 6
    pc = np.random.randint(0, 2, size=(G, T)).astype(np.uint8)
    pp = np.random.randint(0, 2, size=(G, T, P)).astype(np.uint8)
    qc = np.random.randint(0, 2, size=(G, T)).astype(np.uint8)
    qp = np.random.randint(0, 2, size=(G, T, P)).astype(np.uint8)
    pv = np.random.randint(0, 2, size=(B, P)).astype(np.uint8)
    got = O.pi_products_evaluate(PiProducts(pc, pp, qc, qp), pv).to_complex()
    psi = (pc[None] + _ref_parity(pp, pv)) % 2
    phi = (qc[None] + _ref_parity(qp, pv)) % 2
    np.testing.assert_allclose(got, np.prod(np.exp(1j * np.pi * psi * phi), axis=-1), atol=1e-6)


@pytest.mark.parametrize("seed", (0, 42))
def test_phase_pairs_closed_form(seed):
    np.random.seed(seed)
    G, T, P, B = 3, 3, 5, 6
    al = np.random.randint(0, 8, size=(G, T)).astype(np.uint8)
    ap = np.random.randint(0, 2, size=(G, T, P)).astype(np.uint8)
    be = np.random.randint(0, 8, size=(G, T)).astype(np.uint8)
    bp = np.random.randint(0, 2, size=(G, T, P)).astype(np.uint8)
    counts = np.array([T, T - 1, 0], dtype=np.int32)
    pv = np.random.randint(0, 2, size=(B, P)).astype(np.uint8)
    got = O.phase_pairs_evaluate(PhasePairs(al, ap, be, bp, counts), pv).to_complex()
    ea = np.exp(1j * np.pi * al[None] / 4 + 1j * np.pi * _ref_parity(ap, pv))
    eb = np.exp(1j * np.pi * be[None] / 4 + 1j * np.pi * _ref_parity(bp, pv))
    term = 1 + ea + eb - ea * eb
    mask = np.arange(T)[None, :] < counts[:, None]
    np.testing.assert_allclose(got, np.prod(np.where(mask[None], term, 1.0), axis=-1), atol=1e-5)


# ---- evaluate edge cases ---------------------------------------------------------


def test_evaluate_empty_returns_zero():
    """unit/compile/test_compile.py:31-46."""
    z = O.evaluate(empty_scalar_graphs(0), np.zeros((5, 0), np.uint8))
    assert z.shape == (5,) and np.array_equal(z, np.zeros(5, complex))
    z = O.evaluate(empty_scalar_graphs(2), np.ones((3, 2), np.uint8))
    assert z.shape == (3,) and np.array_equal(z, np.zeros(3, complex))


def test_noisy_t_component_is_a_probability():
    """A physically consistent 2-term program: marginals sum to the normalisation."""
    comp = synth.noisy_t_component(0, 0)
    lv0, lv1 = comp.compiled_scalar_graphs
    for fbit in (0, 1):
        p = [abs(O.evaluate(lv1, np.array([[fbit, m]], np.uint8))[0]) for m in (0, 1)]
        n = abs(O.evaluate(lv0, np.array([[fbit]], np.uint8))[0])
        assert abs(p[0] + p[1] - n) < 1e-6
        assert abs(p[1 ^ fbit] - np.sin(np.pi / 8) ** 2) < 1e-6
