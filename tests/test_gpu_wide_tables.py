"""Pattern tables in front of the sparse-column kernel (wide components, more than 64 parameters): the first pass
k_sample_lw<true> finishes the rows whose selected f bits have weight <= depth from tables indexed by the colex rank
of the pattern (built on the device by unranking), the others go through row lists to k_sample4w and its overflow to
the row kernel.  Results must not depend on the depth, the launch API or the shard; every case against the oracle."""

import warnings

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import synth

pytestmark = pytest.mark.gpu


def wide_program(seed=0):
    comps = [dict(n=3, F=200, G=[1, 2, 2, 3], density=0.08), dict(n=2, F=90, G=[2, 3, 4], density=0.1)]
    return synth.physical_program(num_f=320, n_direct=100, components=comps, seed=seed)


@pytest.mark.parametrize("cap", [0, 1, 2, 3, 4])
def test_every_depth_matches_oracle(hip, cap):
    prog = wide_program(11)
    orc = OC.OracleProgram(prog)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = hip.HipProgram(prog, pattern_tables=cap)
        info = hp.info()
        assert info["wide_sparse_kernel"] and info["pattern_tables"] and info["pattern_max_weight"] == [cap, cap]
        for p_bit in (0.0, 0.004, 0.015, 0.05):
            f = synth.synth_f(3000, 320, p_bit, seed=int(p_bit * 1000) + cap)
            want, wdev = orc.sample_program(f, (cap, 6), return_devs=True)
            got, gdev = hp.sample_batch(f, (cap, 6))
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
        hp.close()  # the weight-4 table of 200 bits is 2.1 GB


@pytest.mark.parametrize("deepen", [False, True])
def test_depth_on_demand_and_bits_do_not_change(hip, deepen, monkeypatch):
    """One wide component (k_sample_wide): the weight-4 table of a 200-bit component is 2.1 GB and 24 ms of build for 10 % of
    rate, so a handle deepens only after `deep_after` rows that miss a fifth of the time (2e10 by default: not here) or at once
    on request (TSIM_AMD_DEEP_TABLES=1, after three such launches).  Same bits either way."""
    prog, cfg = synth.config_program("C5")
    orc = OC.OracleProgram(prog)
    if deepen:
        monkeypatch.setenv("TSIM_AMD_DEEP_TABLES", "1")
    hp = hip.HipProgram(prog)
    assert hp.info()["pattern_tables"] and hp.info()["pattern_max_weight"] == [3]
    B = 20_000
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(8):
            f = synth.synth_f(B, cfg["num_f"], cfg["p_bit"], seed=300 + i)
            got, _ = hp.sample_batch(f, (i, 2))
            np.testing.assert_array_equal(got, orc.sample_program(f, (i, 2)))
        assert hp.info()["pattern_max_weight"] == ([4] if deepen else [3])
        if deepen:
            assert hp.info()["pattern_table_bytes"] > 1 << 30
        for i in range(3):
            f = synth.synth_f(B, cfg["num_f"], 0.003, seed=400 + i)
            got, _ = hp.sample_batch(f, (i, 3), shot_offset=7 * i)
            np.testing.assert_array_equal(got, orc.sample_program(f, (i, 3), shot_offset=7 * i))
    hp.close()
