"""Every switch the launch planner still has: ``tsim_tune_keys`` lists the TSIM_AMD_TUNE keys, this file
holds a non-default value for each, and a C2 / C4 / C5 slice through ``tsim_sample_steps_device`` must equal the C oracle
under every one of them - and under the public environment switches.  A key added to the library without a line here fails
the first test."""

import os

import numpy as np
import pytest

from oracle import oracle_c as OC
from test_gpu_steps import _run_steps, _subkeys
from tsim_amd import _lib, prng, synth

pytestmark = pytest.mark.gpu

# key -> the non-default values worth running (the default is what every other test runs)
TUNE = {
    "defer_hard": ["0"], "defer_group": ["1", "8"], "lw_fast": ["0"], "wide_fused": ["0"], "wide_compact": ["0"], "wide_tables": ["0"],
    "wide_depth": ["3"], "wide_passes": ["1"], "hard_wave": ["0"], "hard_wave_rows": ["0", "100000"], "hard_inline_rows": ["0"],
    "hard_comp_par": ["0"], "hard_overflow": ["0"], "deep_after": ["1"], "fused_lanes": ["1", "4"], "fused_max": ["3", "16"],
    "gen": ["0", "2"], "trie": ["0", "2"], "shallow": ["0"], "x3": ["0"], "x4": ["0", "1"], "noise_wave": ["0"], "noise_fused": ["0"],
}
ENV = [{"TSIM_AMD_ADAPTIVE": "0"}, {"TSIM_AMD_FUSED_STEPS": "0"}, {"TSIM_AMD_DEEP_TABLES": "1"}, {"TSIM_AMD_DEEP_TABLES": "-1"},
       {"TSIM_AMD_MODE": "faithful"}, {"TSIM_AMD_MODE": "strict"}, {"TSIM_AMD_KERNEL": "rows"}, {"TSIM_AMD_PATTERN_TABLES": "0"}, {"TSIM_AMD_PATTERN_TABLE_MB": "2"}, {"TSIM_AMD_TABLE_BUILD": "rows"}]


def test_every_tune_key_has_a_case():
    keys = set(_lib.load().tsim_tune_keys().decode().split(","))
    assert keys == set(TUNE), keys ^ set(TUNE)
    assert len(keys) <= 24  # (round 5: 39 knobs.* names in csrc/; what is left must stay countable)


_WANT = {}


def _slice(hip, cn, env):
    prog, cfg = synth.config_program(cn)
    nf, B, n = cfg["num_f"], 3000, 3
    fs = [synth.synth_f(B, nf, cfg["p_bit"] * (1 + i), seed=300 + i) for i in range(n)]
    key = prng.key(77)
    if cn not in _WANT:
        op = OC.OracleProgram(prog)
        _, subs = _subkeys(key, n)
        _WANT[cn] = [np.packbits(op.sample_program(fs[i], subs[i]), axis=1, bitorder="little") for i in range(n)]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        hp = hip.HipProgram(prog)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    _run_steps(hp, prog, fs[:2], prng.key(1), nf, packed=True)  # launch-plan feedback
    outs, _ = _run_steps(hp, prog, fs, key, nf, packed=True)
    hp.close()
    for i in range(n):
        np.testing.assert_array_equal(outs[i], _WANT[cn][i], err_msg=f"{cn} batch {i} under {env}")


@pytest.mark.parametrize("key,val", [(k, v) for k, vs in TUNE.items() for v in vs])
def test_oracle_slice_under_each_tune_value(hip, key, val):
    for cn in ("C2", "C4", "C5"):
        _slice(hip, cn, {"TSIM_AMD_TUNE": f"{key}={val}"})


@pytest.mark.parametrize("env", ENV, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_oracle_slice_under_each_public_switch(hip, env):
    for cn in ("C2", "C4", "C5"):
        _slice(hip, cn, env)
