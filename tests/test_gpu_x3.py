"""Narrow components of 65..128 parameters: since round 5 the chunk-table kernels hold x in three words (at most 64 selected
f bits + outcome bits up to bit 79: NCH = 20) or four (up to 128 parameters, f_sel itself beyond 64 bits: NCH = 32, first pass
k_sample_gen, tables to weight 4) - tsim_kernel4.hip.h sample4_block, tsim_kernel4h.hip.h - so such a program keeps the narrow
family - fused first pass, pattern tables to weight 5+, block-per-row / per-shot hard-row kernels - instead of the wide
path (class F60 of scripts/shape_map.py: one bit past the old wall cost 3.3x).  The reference has no such wall
(src/tsim/sampler.py:28-81 concatenates f_sel and the outcome bits whatever their number).  Every case against the C oracle:
samples and normalisation deviations, bit for bit - through the fused steps API (tables + hard rows), with the tables off
(every row on k_sample4<4, 20>), under dense noise (the per-shot hard-row kernels), and against the wide path (`x3=0`)."""

import os
import warnings

import numpy as np
import pytest

from oracle import oracle_c as OC
from test_gpu_steps import _run_steps, _subkeys
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def _n(n, F, top):
    g = [max(1, int(round(top * (k + 1) / (n + 1)))) for k in range(n + 1)]
    for k in range(1, n + 1):
        g[k] = max(g[k], g[k - 1])
    return dict(n=n, F=F, G=g)


SHAPES = {
    "F60n5": dict(num_f=64, n_direct=4, components=[_n(5, 60, 40)]),                      # 65 parameters (shape class F60)
    "F64n8": dict(num_f=64, n_direct=10, components=[_n(8, 64, 12)]),                     # 72: every f bit selected
    "F64n10": dict(num_f=96, n_direct=20, components=[_n(10, 64, 6)], shuffle_outputs=True),  # 74, ten outputs (k_sample_lw_multi / gen)
    "two": dict(num_f=128, n_direct=30, components=[_n(3, 63, 10), _n(6, 61, 16)], shuffle_outputs=True, direct_flip_fraction=0.2),  # 66 and 67
    "mixed": dict(num_f=64, n_direct=8, components=[_n(2, 20, 8), _n(4, 62, 20)]),        # an ordinary component + a 66-parameter one
    # four words of x (NCH = 32): more than 64 SELECTED bits - first pass k_sample_gen, tables to weight 4 (wide binomials)
    "F70n5": dict(num_f=96, n_direct=4, components=[_n(5, 70, 40)]),                      # 75 parameters (shape class F70)
    "F100n6": dict(num_f=160, n_direct=30, components=[_n(6, 100, 12)], shuffle_outputs=True),  # 106
    "F120n8": dict(num_f=128, n_direct=6, components=[_n(8, 120, 8)]),                    # 128: the last bit of the fourth word
    "bigtwo": dict(num_f=192, n_direct=40, components=[_n(3, 30, 6), _n(4, 90, 16), _n(2, 70, 20)], shuffle_outputs=True, direct_flip_fraction=0.2),
}
BIG = ("F70n5", "F100n6", "F120n8", "bigtwo")


def _program(name, **kw):
    d = dict(SHAPES[name])
    d.update(kw)
    return synth.physical_program(seed=31, **d), d["num_f"]


def _handle(hip, prog, env=None, **kw):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return hip.HipProgram(prog, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("packed", [True, False])
def test_steps_api_equals_oracle(hip, name, packed):
    prog, nf = _program(name)
    orc = OC.OracleProgram(prog)
    hp = _handle(hip, prog)
    info = hp.info()
    assert info["chunk_table_kernel"] and not info["wide_sparse_kernel"], info  # the narrow family took it
    B, n = 2400, 9
    fmax = max(len(c.f_selection) for c in prog.components)
    # mean weight 1 .. 6 of the largest component: tabulated rows, hard rows on the block-per-row kernel, and many hard rows
    fs = [synth.synth_f(B, nf, (1.0 + 2.5 * (i % 3)) / fmax, seed=500 + i) for i in range(n)]
    key = prng.key(77)
    hp.path_counts(reset=True)
    devs = []
    outs, _ = _run_steps(hp, prog, fs, key, nf, packed=packed, calls=[4, 5], devs=devs)
    paths = hp.path_counts()
    assert not ({"sample4w", "lw_lds_wide", "wide", "rows"} & set(paths)), paths
    if name in BIG:
        assert paths.get("gen", 0) >= 1 and not ({"lw_fast", "lw_fastm", "lw_multi", "lw_reg", "lw_lds", "lw_fast1"} & set(paths)), paths  # f_sel beyond 64 bits: k_sample_gen only
    _, subs = _subkeys(key, n)
    for i in range(n):
        want, wdev = orc.sample_program(fs[i], subs[i], return_devs=True)
        np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"{name} batch {i} ({paths})")
        np.testing.assert_array_equal(devs[i][: len(prog.components)], np.asarray(wdev, np.float32), err_msg=f"{name} batch {i}: deviations")
    hp.close()


@pytest.mark.parametrize("name", ["F60n5", "F64n8", "two", "F70n5", "F120n8", "bigtwo"])
@pytest.mark.parametrize("mode", ["tables_off", "dense", "approx"])
def test_every_row_on_the_chunk_table_kernels(hip, name, mode):
    """Tables off: k_sample4<4, 20> on every row (and its normalisation-check block).  Dense noise: most rows hard - the
    first pass hands them to k_sample4h / k_sample4h_multi or the launch plan skips the tables.  approx: the float branch."""
    prog, nf = _program(name, approx=(mode == "approx"))
    orc = OC.OracleProgram(prog)
    fmax = max(len(c.f_selection) for c in prog.components)
    p_bit = {"tables_off": 2.0 / fmax, "dense": 0.25, "approx": 3.0 / fmax}[mode]
    hp = _handle(hip, prog, env={"TSIM_AMD_PATTERN_TABLES": "0"} if mode == "tables_off" else None)
    for i in range(4):
        f = synth.synth_f(1800, nf, p_bit, seed=60 + i)
        want, wdev = orc.sample_program(f, (i, 8), return_devs=True)
        got, gdev = hp.sample_batch(f, (i, 8))
        np.testing.assert_array_equal(got, want, err_msg=f"{name} {mode} call {i}")
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
    # shards: the second one runs no normalisation check
    f = synth.synth_f(2000, nf, p_bit, seed=3)
    want = orc.sample_program(f, (5, 5))
    a, _ = hp.sample_batch(f[:777], (5, 5))
    b, _ = hp.sample_batch(f[777:], (5, 5), shot_offset=777)
    np.testing.assert_array_equal(np.concatenate([a, b]), want)
    hp.close()


@pytest.mark.parametrize("name,off", [("F60n5", "x3=0"), ("F70n5", "x4=0")])
def test_three_and_four_words_equal_the_wide_path(hip, name, off):
    """The same batches with x in three / four words and on the round-4 path (`x3=0` / `x4=0`: the component is 'wide'):
    identical rows, and each handle really took its path."""
    prog, nf = _program(name)
    fs = [synth.synth_f(5000, nf, (1.0 + i) / 60, seed=9 + i) for i in range(4)]
    key = prng.key(4)
    res = []
    for tune in (None, off):
        hp = _handle(hip, prog, env={"TSIM_AMD_TUNE": tune} if tune else None)
        info = hp.info()
        hp.path_counts(reset=True)
        outs, _ = _run_steps(hp, prog, fs, key, nf, packed=True)
        res.append((outs, hp.path_counts(), info["chunk_table_kernel"]))
        hp.close()
    assert res[0][2] and not res[1][2], (res[0][1], res[1][1])
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["F60n5", "F70n5", "F120n8"])
def test_tiny_batches_and_large_shot_offsets(hip, name):
    """1, 63, 64, 65 rows and shot ranges beyond / across 2^32 through the three- and four-word kernels (serial API: tables for
    F <= 64, every row on k_sample4<4, 32> beyond) and through one fused group of the steps API."""
    prog, nf = _program(name)
    orc = OC.OracleProgram(prog)
    fmax = max(len(c.f_selection) for c in prog.components)
    hp = _handle(hip, prog)
    for B, off in ((1, 0), (63, 0), (64, 0), (65, 0), (130, (1 << 32) + 5), (200, (1 << 32) - 100)):
        f = synth.synth_f(B, nf, 2.0 / fmax, seed=B + 3)
        want = orc.sample_program(f, (B, 4), shot_offset=off)
        got, _ = hp.sample_batch(f, (B, 4), shot_offset=off)
        np.testing.assert_array_equal(got, want, err_msg=f"{name} B {B} shot_offset {off}")
    for B, off in ((65, 0), (200, (1 << 32) + 64)):
        fs = [synth.synth_f(B, nf, 2.0 / fmax, seed=B + i) for i in range(3)]
        key = prng.key(B)
        outs, _ = _run_steps(hp, prog, fs, key, nf, packed=True, shot_offset=off)
        _, subs = _subkeys(key, 3)
        for i in range(3):
            np.testing.assert_array_equal(outs[i], np.packbits(orc.sample_program(fs[i], subs[i], shot_offset=off), axis=1, bitorder="little"),
                                          err_msg=f"{name} steps B {B} shot_offset {off} batch {i}")
    hp.close()
