"""The C oracle (oracle/oracle.c) must agree bit for bit with the numpy oracle and the KATs."""

import numpy as np
import pytest

from conftest import run_batches
from oracle import oracle_c as OC
from oracle import oracle_np as O
from tsim_amd import synth
from tsim_amd.program import CompiledComponent, empty_scalar_graphs, make_program


def c_sample(program, f, key):
    return OC.OracleProgram(program).sample_program(f, key)


def test_c_oracle_kats():
    assert [int(o.sum()) for o in run_batches(c_sample, synth.kat_h_m(), 0, [100] * 4)] == [48, 53, 52, 50]
    assert int(run_batches(c_sample, synth.kat_t_gate(), 0, [100])[0].sum()) == 9
    assert run_batches(c_sample, synth.kat_r_gate(), 0, [10])[0].sum(axis=0).tolist() == [7, 4, 0]
    o = run_batches(c_sample, synth.kat_bell(), 0, [100])[0]
    assert np.array_equal(o[:, 0], o[:, 1]) and int(o[:, 0].sum()) == 48


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C4", "C5"])
def test_c_oracle_matches_numpy_on_baseline_configs(name):
    prog, cfg = synth.config_program(name)
    n = 150 if name == "C4" else 400
    f = synth.synth_f(n, cfg["num_f"], cfg["p_bit"] * 3, seed=7)
    want, wd = O.sample_program(prog, f, (3, 4), return_devs=True)
    got, gd, ov = OC.OracleProgram(prog).sample_program(f, (3, 4), return_devs=True, return_overflow=True)
    if name == "n24x":  # the unconstrained 24-output mixture: the reference's int32 coefficients DO wrap - both restatements wrap alike
        assert ov
    else:
        assert not ov, "reference int32 arithmetic would wrap on this input"
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gd, np.float32), np.asarray(wd, np.float32))


@pytest.mark.parametrize("approx", [False, True])
def test_c_oracle_evaluate_bits(approx):
    rng = np.random.default_rng(11)
    lv = synth.synth_level(rng, 70, 9, approx=approx)  # W = 2 words
    comp = CompiledComponent(tuple(range(70)), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv))
    prog = make_program([comp], [], 70, 0)
    pv = (rng.random((300, 70)) < 0.5).astype(np.uint8)
    z, ex, ov = OC.OracleProgram(prog).evaluate(0, 1, pv, exact=True)
    assert not ov
    want = O.evaluate(lv, pv)
    np.testing.assert_array_equal(z.view(np.float32), want.view(np.float32))
    if not approx:
        coeffs, power = O.evaluate_exact(lv, pv)
        np.testing.assert_array_equal(ex[:, :4], coeffs)
        nz = np.any(coeffs != 0, axis=1)
        np.testing.assert_array_equal(ex[nz, 4], power[nz])


def test_c_oracle_threads_and_shards_agree():
    prog, cfg = synth.config_program("C2")
    f = synth.synth_f(3000, cfg["num_f"], cfg["p_bit"], seed=1)
    op = OC.OracleProgram(prog)
    a = op.sample_program(f, (1, 2), threads=1)
    b = op.sample_program(f, (1, 2), threads=4)
    c = np.concatenate([op.sample_program(f[:1234], (1, 2)), op.sample_program(f[1234:], (1, 2), shot_offset=1234)])
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_overflow_flag_fires():
    """A product that exceeds int32 must raise the oracle's wrap flag."""
    g = dict(A=[(1, []) for _ in range(80)])  # (1 + w)^80: coefficients ~ 2.4^40
    from tsim_amd.program import scalar_graphs_from_terms

    lv = scalar_graphs_from_terms(1, [g])
    comp = CompiledComponent((0,), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv))
    prog = make_program([comp], [], 1, 0)
    _, _, ov = OC.OracleProgram(prog).evaluate(0, 1, np.zeros((1, 1), np.uint8), exact=True)
    assert ov


@pytest.mark.parametrize("name", ["F60", "F70", "F140", "F300", "2wide", "narrow+wide", "n11", "out260_f320", "6narrow_f320",
                                  "n16", "n40", "n24x", "w12", "20narrow", "9wide", "F600"])  # (the last row: round 6, beyond the compiled-in walls)
def test_c_oracle_matches_numpy_on_shape_classes(name):
    """The GPU parity tests of the round-5 shapes (65..128 parameters, several wide components, more than 255 selected
    bits, 11 outputs per component, 260 outputs) use the C oracle as their checker: pin IT to the numpy restatement of the
    reference's data movement on those shapes too (samples and normalisation deviations; small batches - numpy walks
    byte-per-bit arrays)."""
    prog, c = synth.shape_class_program(name)
    f = synth.synth_f(48, c["num_f"], 2.5 * c["p_bit"], seed=13)
    want, wd = O.sample_program(prog, f, (3, 7), return_devs=True)
    got, gd, ov = OC.OracleProgram(prog).sample_program(f, (3, 7), return_devs=True, return_overflow=True)
    if name == "n24x":  # the unconstrained 24-output mixture: the reference's int32 coefficients DO wrap - both restatements wrap alike
        assert ov
    else:
        assert not ov, "reference int32 arithmetic would wrap on this input"
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gd, np.float32), np.asarray(wd, np.float32))
