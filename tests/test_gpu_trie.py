"""Prefix-tree pattern tables (csrc/tsim_trie.hip.h, round 6): components of more than 12 outputs - the reference loops over
any number of levels (src/tsim/sampler.py:62) - against the C oracle, through ``tsim_sample_steps_device``:
the new shape classes, a table budget so small that trees are cut short (rows that reach a missing child are hard rows), the
same bytes with the format switched off, and the format forced onto programs the dense tables serve (``trie=2``)."""

import os

import numpy as np
import pytest

from oracle import oracle_c as OC
from test_gpu_shape_classes import _check_class
from test_gpu_steps import _run_steps, _subkeys
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name", ["n13", "n16", "n24", "n40"])
def test_prefix_trees_serve_the_class(hip, name):
    paths = _check_class(hip, name, B=1500, n=5, packed=True)
    assert paths.get("gen", 0) >= 1, paths


@pytest.mark.parametrize("name,mb", [("n13", 1), ("n13", 4), ("n16", 2), ("n16", 16), ("n24", 1), ("n40", 1), ("n40", 8)])
@pytest.mark.parametrize("packed", [True, False])
def test_trees_cut_short_by_the_budget(hip, name, mb, packed):
    """A budget of 1-8 MB ends the build inside weight 1 or 2: patterns beyond the complete prefix are hard rows."""
    _with_env({"TSIM_AMD_PATTERN_TABLE_MB": str(mb)}, lambda: _check_class(hip, name, B=1300, n=4, packed=packed))


@pytest.mark.parametrize("name", ["n16", "n40"])
def test_format_off_equals_on(hip, name):
    prog, c = synth.shape_class_program(name)
    nf, B, n = c["num_f"], 20_000, 3
    fs = [synth.synth_f(B, nf, 0.03, seed=70 + i) for i in range(n)]
    res = []
    for tune in ("trie=0", "trie=1"):
        def run():
            hp = hip.HipProgram(synth.shape_class_program(name)[0])
            _run_steps(hp, prog, fs[:2], prng.key(1), nf, packed=True, shot_offset=1 << 20)
            hp.path_counts(reset=True)
            outs, _ = _run_steps(hp, prog, fs, prng.key(9), nf, packed=True, shot_offset=1 << 20)
            pc = hp.path_counts()
            hp.close()
            return outs, pc
        res.append(_with_env({"TSIM_AMD_TUNE": tune}, run))
    assert "gen" not in res[0][1] and res[1][1].get("gen", 0) >= 1, (res[0][1], res[1][1])
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["f64", "n1", "n8", "n9", "n11", "3narrow", "6narrow", "6narrow_f320", "n17total", "out121", "F59", "20narrow"])
def test_format_forced_on_dense_classes(hip, name):
    """trie=2: every narrow component's tables as prefix trees (1..11 outputs: root chunks of 1, 2 and 3 levels)."""
    paths = _check_class(hip, name, B=1111, n=4, packed=True, tune="trie=2")
    assert paths.get("gen", 0) >= 1 and not any(k in paths for k in ("lw_fast", "lw_fastm", "lw_multi")), paths


def test_nonsense_marginals_n24x(hip):
    """n24x: int32 coefficients wrap like the reference's (normalisation deviation ~1: the reference raises) - whatever path
    serves it, the bytes are the oracle's."""
    _check_class(hip, "n24x", B=700, n=3, packed=True)


@pytest.mark.parametrize("comps", [[(24, 20), (24, 24)], [(26, 16), (20, 12), (28, 20)], [(13, 10)] * 6])
def test_more_than_forty_compiled_outputs(hip, comps):
    """48, 74 and 78 compiled outputs: the fused groups of k_sample_gen carry fewer batches (its subkey records are a pool of
    batches x outputs); round 5's limit of 40 sent such a program to the row kernel."""
    rng_seed = 10
    cl = []
    for n, F in comps:
        G = [2]
        for _ in range(n):
            G.append(min(6, G[-1] + (1 if len(G) % 5 == 0 else 0)))
        cl.append(dict(n=n, F=F, G=G, density=0.2, shared_delta=0.8))
    nf = 96
    prog = synth.physical_program(num_f=nf, n_direct=9, components=cl, seed=rng_seed)
    hp = hip.HipProgram(prog)
    B, n = 1200, 9
    fs = [synth.synth_f(B, nf, 0.02 * (1 + i % 3), seed=500 + i) for i in range(n)]
    _run_steps(hp, prog, fs[:3], prng.key(1), nf, packed=True)
    import time
    t0 = time.perf_counter()
    while hp.info()["pattern_build_pending"] and time.perf_counter() - t0 < 20.0:
        _run_steps(hp, prog, fs[:1], prng.key(2), nf, packed=True)
    hp.path_counts(reset=True)
    key = prng.key(77)
    devs = []
    outs, _ = _run_steps(hp, prog, fs, key, nf, packed=True, devs=devs)
    paths = hp.path_counts()
    assert paths.get("gen", 0) >= 1, paths
    _, subs = _subkeys(key, n)
    op = OC.OracleProgram(prog)
    for i in range(n):
        want, wdev = op.sample_program(fs[i], subs[i], return_devs=True)
        np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i} ({paths})")
        np.testing.assert_array_equal(devs[i][: len(prog.components)], np.asarray(wdev, np.float32))
    hp.close()


@pytest.mark.parametrize("name", ["n13", "n16", "n24", "n40", "F70"])
@pytest.mark.parametrize("B,shot_offset", [(1, 0), (777, 0), (5000, 0), (3001, 1 << 16)])
def test_one_batch_api_rides_the_general_first_pass(hip, name, B, shot_offset):
    """``tsim_sample_batch`` (the seam of ``backend.sample_program``) on programs whose tables only k_sample_gen reads: the batch
    is a group of one (gen_one in csrc/tsim_sample_gen.hip) instead of every row on the full kernel - same bytes as the oracle,
    padded and bit_packed, with and without the normalisation-check row."""
    prog, c = synth.shape_class_program(name)
    nf = c["num_f"]
    hp = hip.HipProgram(prog)
    op = OC.OracleProgram(prog)
    hp.path_counts(reset=True)
    for i, packed in enumerate([False, True, False]):
        f = synth.synth_f(B, nf, c["p_bit"] * (1 + i), seed=300 + i)
        key = prng.key(50 + i)
        got, gdev = hp.sample_batch(f, key, shot_offset=shot_offset, bit_packed=packed)
        want, wdev = op.sample_program(f, key, return_devs=True, shot_offset=shot_offset)
        if packed:
            want = np.packbits(want, axis=1, bitorder="little")
            got = got[:, : want.shape[1]]
        np.testing.assert_array_equal(np.asarray(got, np.uint8), np.asarray(want, np.uint8), err_msg=f"{name} call {i}")
        if shot_offset == 0:
            np.testing.assert_array_equal(gdev[: len(prog.components)], np.asarray(wdev, np.float32))
    paths = hp.path_counts()
    hp.close()
    assert paths.get("gen", 0) >= 1, paths


@pytest.mark.parametrize("name", ["n16", "n40"])
def test_one_batch_api_general_pass_off_equals_on(hip, name):
    prog, c = synth.shape_class_program(name)
    nf, B = c["num_f"], 30_000
    f = synth.synth_f(B, nf, 0.03, seed=5)
    res = []
    for tune in ("gen=0", "gen=1"):
        def run():
            hp = hip.HipProgram(synth.shape_class_program(name)[0])
            hp.sample_batch(f[:1000], prng.key(1))
            hp.path_counts(reset=True)
            out, dev = hp.sample_batch(f, prng.key(9))
            pc = hp.path_counts()
            hp.close()
            return out, dev, pc
        res.append(_with_env({"TSIM_AMD_TUNE": tune}, run))
    assert "gen" not in res[0][2] and res[1][2].get("gen", 0) >= 1, (res[0][2], res[1][2])
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
