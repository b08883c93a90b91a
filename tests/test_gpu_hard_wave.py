"""The wave-per-row hard-row kernel (k_sample_hw, tsim_kernel_hw.hip.h) against the 64-rows-per-block one
(k_sample4h_multi) and the oracle: pipelined launches whose hard rows it serves - shallow pattern tables so that MANY
rows are hard, the normalisation-check row among them - for exact, approximate and live-padding programs, one and
several components, several graphs per lane (C4: up to 256 graphs in a level)."""

import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def _run(hip, prog, fs, key, nf, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        hp = hip.HipProgram(prog, pattern_tables=1)  # tables to weight 1 only: most noisy rows are hard
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    B = len(fs[0])
    wf, rb = (nf + 63) // 64, (prog.num_outputs + 7) // 8
    d_f = [hp.malloc(B * wf * 8) for _ in fs]
    d_o = [hp.malloc(B * 8 + 16) for _ in fs]
    for d, f in zip(d_f, fs):
        p = np.packbits(f, axis=1, bitorder="little")
        hp.h2d(d, np.ascontiguousarray(np.pad(p, ((0, 0), (0, wf * 8 - p.shape[1])))))
    outs = []
    for rep in range(2):  # the first call gives the launch plan its feedback; the second one is fused and deferred
        ks = (C.c_uint32 * 2)(key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF)
        hp.sample_steps_device([d.ptr for d in d_f], B, nf, ks, [d.ptr for d in d_o], out_bit_packed=True)
        hp.synchronize()
    for d in d_o:
        got = np.zeros((B, rb), np.uint8)
        hp.d2h(got, d)
        outs.append(got)
    hp.close()
    return outs


@pytest.mark.parametrize("name,kw", [("C2", {}), ("C2", dict(approx=True)), ("C2", dict(live_padding=True, approx=True)), ("C3", {}),
                                      ("C4", {}), ("C4", dict(approx=True))])
def test_hard_rows_wave_per_row_equals_block_kernel_and_oracle(hip, name, kw):
    prog, cfg = synth.config_program(name, **kw)
    nf, B, n = cfg["num_f"], 4000, 6
    fs = [synth.synth_f(B, nf, 0.03, seed=70 + i) for i in range(n)]
    key = prng.key(9)
    a = _run(hip, prog, fs, key, nf, {"TSIM_AMD_TUNE": "hard_wave=1"})
    b = _run(hip, prog, fs, key, nf, {"TSIM_AMD_TUNE": "hard_wave=0"})
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    op = OC.OracleProgram(prog)
    k = key
    for i in range(n):
        k, sub = prng.split(k)
        if i in (0, n - 1):
            want = op.sample_program(fs[i], sub)
            np.testing.assert_array_equal(a[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i}")
