"""The benchmark's workload variants against the oracle: approximate floatfactors (the float32 sum branch,
compile/evaluate.py:56-59) and live padding (terms the packer's algebra cannot cancel: complex amplitudes, full-rank
quadratic forms) - through the pattern-table pipeline and through the full kernel alone."""

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["C2", "C3"])
@pytest.mark.parametrize("kw", [dict(live_padding=True), dict(live_padding=True, approx=True), dict(approx=True)])
@pytest.mark.parametrize("tables", [True, False])
def test_variant_samples_equal_oracle(hip, name, kw, tables):
    prog, cfg = synth.config_program(name, **kw)
    B = 2500
    f = synth.synth_f(B, cfg["num_f"], 0.04, seed=11)
    key = prng.key(2024)
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, key, return_devs=True, return_overflow=True)
    assert not ov
    hp = hip.HipProgram(prog, pattern_tables=tables)
    got, gdev = hp.sample_batch(f, key)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
    info = hp.info()
    if kw.get("live_padding"):
        ref_rows = hip.HipProgram(synth.config_program(name)[0], pattern_tables=False).info()["total_rows"]
        assert info["total_rows"] > 1.5 * ref_rows  # the padding survived the pack-time algebra
    hp.close()


def test_live_padding_amplitudes_are_complex_and_equal_the_oracle(hip):
    prog, cfg = synth.config_program("C2", live_padding=True)
    comp = prog.components[0]
    lv = comp.compiled_scalar_graphs[3]
    rng = np.random.default_rng(3)
    pv = (rng.random((300, lv.n_params)) < 0.2).astype(np.uint8)
    hp = hip.HipProgram(prog)
    z, ex = hp.evaluate(0, 3, pv, exact=True)
    wz, wex, ov = OC.OracleProgram(prog).evaluate(0, 3, pv, exact=True)
    assert not ov
    np.testing.assert_array_equal(ex, wex)
    np.testing.assert_array_equal(z.view(np.float32), wz.view(np.float32))
    assert (np.abs(z.imag) > 1e-9).any()
