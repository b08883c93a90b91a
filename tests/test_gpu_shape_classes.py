"""Every shape class of ``synth.SHAPE_CLASSES`` (one eligibility wall of the kernels moved on its own from the nearest
BASELINE configuration: f-row width, output count, number / mix of components, outputs per component, selected bits per
component) through ``tsim_sample_steps_device`` against the C oracle, bit for bit, with the kernel family that served it
recorded.  The reference takes every shape through one code path (src/tsim/sampler.py:117-167, compile/pipeline.py:55-102);
here the shape picks the kernel, so every class needs its own parity test."""

import os

import numpy as np
import pytest

from oracle import oracle_c as OC
from test_gpu_steps import _run_steps, _subkeys
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu

# classes the general fused first pass (k_sample_gen, tsim_gen.hip.h) must serve by default
GEN_CLASSES = ["f160", "f320", "f600", "out65", "out121", "out121_f320", "out260_f320", "6narrow_f320", "n17total"]


def _check_class(hip, name, B, n, packed, tune=None):
    prog, c = synth.shape_class_program(name)
    nf = c["num_f"]
    old = os.environ.get("TSIM_AMD_TUNE")
    if tune is not None:
        os.environ["TSIM_AMD_TUNE"] = tune
    try:
        hp = hip.HipProgram(prog)
    finally:
        if tune is not None:
            if old is None:
                os.environ.pop("TSIM_AMD_TUNE", None)
            else:
                os.environ["TSIM_AMD_TUNE"] = old
    warm = [synth.synth_f(B, nf, c["p_bit"], seed=900 + i) for i in range(3)]
    _run_steps(hp, prog, warm, prng.key(1), nf, packed=packed)  # launch-plan feedback
    # a fresh handle samples with shallow tables while its default depth is built in the background: which kernels serve the
    # batches below must not depend on whether that build has landed yet (n8: 237 MB, ~10 ms) - wait for it
    import time
    t0 = time.perf_counter()
    while hp.info()["pattern_build_pending"] and time.perf_counter() - t0 < 20.0:
        _run_steps(hp, prog, warm[:1], prng.key(2), nf, packed=packed)
    _run_steps(hp, prog, warm, prng.key(3), nf, packed=packed)  # ... and the plan's feedback on the tables now in place
    hp.path_counts(reset=True)
    # noise levels around the nominal one: weight > table depth rows (hard rows) occur in every batch at 3 x p_bit
    fs = [synth.synth_f(B, nf, c["p_bit"] * (1 + (i % 3)), seed=40 + i) for i in range(n)]
    key = prng.key(123)
    devs = []
    outs, key_after = _run_steps(hp, prog, fs, key, nf, packed=packed, devs=devs)
    paths = hp.path_counts()
    end_key, subs = _subkeys(key, n)
    assert key_after == (end_key[0] & 0xFFFFFFFF, end_key[1] & 0xFFFFFFFF)
    op = OC.OracleProgram(prog)
    for i in range(n):
        want, wdev = op.sample_program(fs[i], subs[i], return_devs=True)
        np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"{name} batch {i} ({paths})")
        np.testing.assert_array_equal(devs[i][: len(prog.components)], np.asarray(wdev, np.float32), err_msg=f"{name} batch {i}: normalisation deviations")
    hp.close()
    return paths


@pytest.mark.parametrize("name", list(synth.SHAPE_CLASSES))
def test_class_equals_oracle(hip, name):
    paths = _check_class(hip, name, B=1500, n=5, packed=True)
    if name in GEN_CLASSES:
        assert paths.get("gen", 0) >= 1, paths


@pytest.mark.parametrize("name", ["f64", "f128", "out64", "3narrow", "6narrow", "n1", "n8", "n9", "n11", "F16", "F59"] + GEN_CLASSES)
@pytest.mark.parametrize("packed", [True, False])
def test_general_first_pass_forced(hip, name, packed):
    """TSIM_AMD_TUNE=gen=2: k_sample_gen also where a register first pass applies; ragged batch (not a multiple of 64),
    padded and bit_packed rows."""
    paths = _check_class(hip, name, B=1111, n=4, packed=packed, tune="gen=2")
    assert paths.get("gen", 0) >= 1 and not any(k in paths for k in ("lw_fast", "lw_fastm", "lw_multi")), paths


def test_general_first_pass_off_equals_on(hip):
    """gen=0 (the round-4 routing) and the default give the same bytes on a class the new kernel serves, at a size with
    several row blocks per batch and a non-zero shot offset."""
    import ctypes as C  # noqa: F401

    name, B, n = "out121_f320", 70_000, 3
    prog, c = synth.shape_class_program(name)
    nf = c["num_f"]
    fs = [synth.synth_f(B, nf, 0.03, seed=70 + i) for i in range(n)]
    res = []
    for tune in ("gen=0", "gen=1"):
        os.environ["TSIM_AMD_TUNE"] = tune
        try:
            hp = hip.HipProgram(synth.shape_class_program(name)[0])
        finally:
            os.environ.pop("TSIM_AMD_TUNE", None)
        _run_steps(hp, prog, fs[:2], prng.key(1), nf, packed=True, shot_offset=1 << 20)
        hp.path_counts(reset=True)
        outs, _ = _run_steps(hp, prog, fs, prng.key(9), nf, packed=True, shot_offset=1 << 20)
        res.append((outs, hp.path_counts()))
        hp.close()
    assert "gen" not in res[0][1] and res[1][1].get("gen", 0) >= 1, (res[0][1], res[1][1])
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(a, b)
