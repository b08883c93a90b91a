"""Device noise drawn in the same call as the sampling (``tsim_sample_steps_noise_device``, round 6): one-component programs over
narrow rows run noise + first pass as ONE kernel (csrc/tsim_noise_fused.hip.h), every other program the noise kernel in front of
its own first pass.  Whatever the path: the f rows are the bytes ``tsim_noise_sample_device`` writes for the same keys, and the
outputs are the C oracle's on those rows (reference: src/tsim/sampler.py:393-400 - channel sampler, then sample_program)."""

import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import prng, synth
from tsim_amd.channels import ChannelSampler, error_probs, pauli_channel_1_probs

pytestmark = pytest.mark.gpu


def _run(hip, prog, nf, probs, T, B, n, *, fused_call, tune=None, packed=True, shot_offset=0):
    old = os.environ.get("TSIM_AMD_TUNE")
    if tune is not None:
        os.environ["TSIM_AMD_TUNE"] = tune
    try:
        hp = hip.HipProgram(prog)
        dn = hip.DeviceNoiseSampler(hp, ChannelSampler(probs, T, seed=5))
    finally:
        if tune is not None:
            if old is None:
                os.environ.pop("TSIM_AMD_TUNE", None)
            else:
                os.environ["TSIM_AMD_TUNE"] = old
    WF, WO, RB = max(1, (nf + 63) // 64), (prog.num_outputs + 63) // 64, (prog.num_outputs + 7) // 8
    d_f = [hp.malloc(max(16, B * WF * 8)) for _ in range(n)]
    d_o = [hp.malloc(max(16, B * max(RB, 8 * WO))) for _ in range(n)]
    key, nkey = prng.key(21), prng.key(99)
    ks = (C.c_uint32 * 2)(key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF)
    nks = (C.c_uint32 * 2)(nkey[0] & 0xFFFFFFFF, nkey[1] & 0xFFFFFFFF)
    for rep in range(2):  # twice: launch-plan feedback, then the tables in place
        ks[0], ks[1] = key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF
        nks[0], nks[1] = nkey[0] & 0xFFFFFFFF, nkey[1] & 0xFFFFFFFF
        hp.path_counts(reset=True)
        if fused_call:
            hp.sample_steps_noise_device(dn, [d.ptr for d in d_f], B, nf, ks, nks, [d.ptr for d in d_o], shot_offset=shot_offset, out_bit_packed=packed)
        else:  # by hand: the same key chains
            kn = nkey
            for j in range(n):
                kn, sub = prng.split(kn)
                dn.sample_into(d_f[j].ptr, B, sub)
            hp.sample_steps_device([d.ptr for d in d_f], B, nf, ks, [d.ptr for d in d_o], shot_offset=shot_offset, out_bit_packed=packed)
        hp.synchronize()
    paths = hp.path_counts()
    fs, outs = [], []
    for j in range(n):
        f = np.zeros((B, WF * 8), np.uint8)
        hp.d2h(f, d_f[j])
        fs.append(f)
        if packed:
            o = np.zeros((B, RB), np.uint8)
            hp.d2h(o, d_o[j])
        else:
            raw = np.zeros((B, WO * 8), np.uint8)
            hp.d2h(raw, d_o[j])
            o = np.packbits(np.unpackbits(raw, axis=1, bitorder="little")[:, : prog.num_outputs], axis=1, bitorder="little")
        outs.append(o)
    hp.close()
    return fs, outs, paths, (int(ks[0]), int(ks[1])), (int(nks[0]), int(nks[1]))


def _probs(nf, p):
    return [error_probs(p)] * nf, np.eye(nf, dtype=np.uint8)


@pytest.mark.parametrize("cn,B,n", [("C2", 20000, 5), ("C2", 4096, 3), ("C2", 5000, 9), ("C2", 100, 2), ("C3", 9000, 4)])
@pytest.mark.parametrize("packed", [True, False])
def test_fused_noise_equals_the_two_kernels_and_the_oracle(hip, cn, B, n, packed):
    prog, cfg = synth.config_program(cn)
    nf = cfg["num_f"]
    probs, T = _probs(nf, cfg["p_bit"])
    a = _run(hip, prog, nf, probs, T, B, n, fused_call=True, packed=packed)
    b = _run(hip, prog, nf, probs, T, B, n, fused_call=True, tune="noise_fused=0", packed=packed)
    c = _run(hip, prog, nf, probs, T, B, n, fused_call=False, packed=packed)
    assert a[2].get("noise_fast", 0) >= 1, a[2]
    assert "noise_fast" not in b[2] and "noise_fast" not in c[2]
    assert a[3] == b[3] == c[3] and a[4] == b[4]  # both key chains advanced alike
    for j in range(n):
        np.testing.assert_array_equal(a[0][j], c[0][j], err_msg=f"f rows of batch {j}")
        np.testing.assert_array_equal(b[0][j], c[0][j])
        np.testing.assert_array_equal(a[1][j], c[1][j], err_msg=f"outputs of batch {j}")
        np.testing.assert_array_equal(b[1][j], c[1][j])
    # ... and the oracle on the downloaded rows
    op = OC.OracleProgram(prog)
    k = prng.key(21)
    for j in range(n):
        k, sub = prng.split(k)
        if j in (0, n - 1):
            f = np.unpackbits(a[0][j], axis=1, bitorder="little")[:, :nf]
            assert 0.5 * cfg["p_bit"] < f.mean() < 1.5 * cfg["p_bit"] or B < 1000
            want = np.packbits(op.sample_program(f, sub), axis=1, bitorder="little")
            np.testing.assert_array_equal(a[1][j], want, err_msg=f"batch {j} against the oracle")


@pytest.mark.parametrize("cn", ["C4", "C5"])
def test_other_first_passes_get_the_noise_kernel_in_front(hip, cn):
    """C4 (three components: k_sample_lw_fastm) and C5 (k_sample_wide): same call, the noise kernel per batch on the lane."""
    prog, cfg = synth.config_program(cn)
    nf = cfg["num_f"]
    probs, T = _probs(nf, cfg["p_bit"])
    a = _run(hip, prog, nf, probs, T, 6000, 4, fused_call=True)
    c = _run(hip, prog, nf, probs, T, 6000, 4, fused_call=False)
    assert "noise_fast" not in a[2]
    for j in range(4):
        np.testing.assert_array_equal(a[0][j], c[0][j])
        np.testing.assert_array_equal(a[1][j], c[1][j])
    op = OC.OracleProgram(prog)
    _, sub = prng.split(prng.key(21))
    f = np.unpackbits(a[0][0], axis=1, bitorder="little")[:, :nf]
    np.testing.assert_array_equal(a[1][0], np.packbits(op.sample_program(f, sub), axis=1, bitorder="little"))


def test_general_first_pass_and_multi_outcome_channels(hip):
    """A class served by k_sample_gen (320-bit rows) with Pauli channels (three outcomes each) through an error transform."""
    prog, c = synth.shape_class_program("f320")
    nf = c["num_f"]
    rng = np.random.default_rng(4)
    probs = [pauli_channel_1_probs(0.004, 0.003, 0.005)] * 200
    T = (rng.random((nf, 400)) < 0.01).astype(np.uint8)
    a = _run(hip, prog, nf, probs, T, 7000, 3, fused_call=True)
    c2 = _run(hip, prog, nf, probs, T, 7000, 3, fused_call=False)
    for j in range(3):
        np.testing.assert_array_equal(a[0][j], c2[0][j])
        np.testing.assert_array_equal(a[1][j], c2[1][j])


def test_fused_noise_with_a_shot_offset_and_dense_noise(hip):
    """shot_offset != 0 (a shard: no normalisation check) and a noise level at which most rows are hard (the plan leaves the
    tables: the one-batch path gets the noise kernel in front)."""
    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    for p_bit, off in ((0.02, 1 << 20), (0.3, 0)):
        probs, T = _probs(nf, p_bit)
        a = _run(hip, prog, nf, probs, T, 8192, 4, fused_call=True, shot_offset=off)
        c = _run(hip, prog, nf, probs, T, 8192, 4, fused_call=False, shot_offset=off)
        for j in range(4):
            np.testing.assert_array_equal(a[0][j], c[0][j])
            np.testing.assert_array_equal(a[1][j], c[1][j])
