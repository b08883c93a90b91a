"""k_sample_wide (tsim_wide.hip.h): programs with ONE wide component (more than 64 parameters) in a single kernel -
pattern tables for the light rows, the sparse-column evaluation of the rows they miss from a per-wave LDS queue, a
generic pass for rows heavier than a dense pass takes and for the normalisation check.  Every case against the oracle
(reference: src/tsim/sampler.py:28-167), across table depths, noise levels (all three row classes), output layouts
(bits spread over several words, flips, shuffled columns), shards, the serial and the several-batches API."""

import ctypes as C
import os
import warnings

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def one_wide(seed, *, n=3, F=200, G=(1, 2, 2, 3), num_f=320, n_direct=118, shuffle=False, flips=0.0, identity=True, density=0.08,
             live=False, **terms):
    comps = [dict(n=n, F=F, G=list(G), density=density, **terms)]
    return synth.physical_program(num_f=num_f, n_direct=n_direct, components=comps, seed=seed, shuffle_outputs=shuffle,
                                  direct_flip_fraction=flips, identity_direct=identity, live_padding=live)


def _packed(f, wf):
    p = np.packbits(f, axis=1, bitorder="little")
    return np.ascontiguousarray(np.pad(p, ((0, 0), (0, wf * 8 - p.shape[1]))))


def _steps(hp, prog, fs, key, nf, *, packed, shot_offset=0, calls=None, devs=False):
    B = len(fs[0])
    wf, wo, rb = (nf + 63) // 64, (prog.num_outputs + 63) // 64, (prog.num_outputs + 7) // 8
    d_f = [hp.malloc(B * wf * 8) for _ in fs]
    d_o = [hp.malloc(max(B * wo * 8, 16)) for _ in fs]
    d_d = [hp.malloc(16) for _ in fs]
    for d, f in zip(d_f, fs):
        hp.h2d(d, _packed(f, wf))
    ks = (C.c_uint32 * 2)(key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF)
    i = 0
    for n in (calls or [len(fs)]):
        hp.sample_steps_device([d.ptr for d in d_f[i:i + n]], B, nf, ks, [d.ptr for d in d_o[i:i + n]], shot_offset=shot_offset,
                               out_bit_packed=packed, d_norm_dev=[d.ptr for d in d_d[i:i + n]] if devs else None)
        i += n
    hp.synchronize()
    outs, dv = [], []
    for d, dd in zip(d_o, d_d):
        if packed:
            got = np.zeros((B, rb), np.uint8)
            hp.d2h(got, d)
        else:
            raw = np.zeros((B, wo * 8), np.uint8)
            hp.d2h(raw, d)
            got = np.packbits(np.unpackbits(raw, axis=1, bitorder="little")[:, : prog.num_outputs], axis=1, bitorder="little")
        outs.append(got)
        one = np.zeros(1, np.float32)
        hp.d2h(one, dd)
        dv.append(float(one[0]))
    for d in d_f + d_o + d_d:
        d.free()
    return outs, dv


def _subkeys(key, n):
    subs = []
    for _ in range(n):
        key, sub = prng.split(key)
        subs.append(sub)
    return subs


@pytest.mark.parametrize("cap", [0, 2, 3, 4])
@pytest.mark.parametrize("p_bit", [0.0, 0.004, 0.02, 0.05, 0.12])
def test_serial_api_every_depth_and_noise_level(hip, cap, p_bit):
    """hits only (p_bit 0), hits + misses, and - 0.05 / 0.12: mean weight 10 / 24 - mostly heavy rows through the generic pass."""
    prog = one_wide(21)
    orc = OC.OracleProgram(prog)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = hip.HipProgram(prog, pattern_tables=cap)
        f = synth.synth_f(3000, 320, p_bit, seed=int(p_bit * 1000) + cap)
        want, wdev = orc.sample_program(f, (cap, 9), return_devs=True)
        got, gdev = hp.sample_batch(f, (cap, 9))
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
        a, _ = hp.sample_batch(f[:1111], (cap, 9), bit_packed=True)
        b, _ = hp.sample_batch(f[1111:], (cap, 9), shot_offset=1111, bit_packed=True)
        np.testing.assert_array_equal(np.concatenate([a, b]), np.packbits(want, axis=1, bitorder="little"))
        hp.close()


@pytest.mark.parametrize("shape", [
    dict(n=1, F=70, G=(2, 3), num_f=128, n_direct=20),
    dict(n=2, F=255, G=(1, 2, 4), num_f=400, n_direct=90, shuffle=True, flips=0.3),
    dict(n=5, F=120, G=(2, 3, 4, 4, 5, 6), num_f=192, n_direct=100, shuffle=True, identity=False, flips=0.2),
    dict(n=8, F=100, G=(1, 1, 2, 2, 2, 3, 3, 3, 4), num_f=256, n_direct=180, shuffle=True, identity=False),
    dict(n=3, F=200, G=(1, 2, 2, 3), num_f=512, n_direct=10, shuffle=True),
])
@pytest.mark.parametrize("packed", [True, False])
def test_shapes_through_the_steps_api(hip, shape, packed):
    """Other widths of f rows and output rows, 1 to 8 outputs scattered over the output words, non-identity direct tables with
    flips; several batches per call (groups of up to 8 in one grid), split calls, against the oracle batch by batch."""
    prog = one_wide(5, **shape)
    nf = shape["num_f"]
    # (bit_packed rows of 3, 14, 23 ... bytes start at odd addresses: k_sample_wide writes them with unaligned dwords and
    # tail bytes since round 5 - the assertion on `launches` below proves these shapes no longer leave the kernel)
    orc = OC.OracleProgram(prog)
    B, n = 2500, 11
    p_mean = 4.0 / shape["F"]
    fs = [synth.synth_f(B, nf, p_mean * (0.5 + (i % 4)), seed=40 + i) for i in range(n)]
    key = prng.key(31)
    subs = _subkeys(key, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = hip.HipProgram(prog)
        hp.profile_set_sampling(1)
        hp.profile_enable(2)
        hp.profile_read(reset=True)
        hp.profile_read_steps()
        outs, devs = _steps(hp, prog, fs, key, nf, packed=packed, calls=[3, 8], devs=True)
        _, launches = hp.profile_read(reset=True)
        assert hp.profile_read_steps() == n and launches == 2, "the fused wide kernel did not take the groups"
        hp.profile_enable(False)
        for i in range(n):
            want, wdev = orc.sample_program(fs[i], subs[i], return_devs=True)
            np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i}")
            assert np.float32(devs[i]) == np.float32(wdev[0]), f"normalisation deviation of batch {i}"
        hp.close()


SHARED_SHAPES = [
    dict(),                                                       # C5's shape: 8 graphs, 76 parity bits
    dict(n=1, F=70, G=(2, 3), num_f=128, n_direct=20),
    dict(n=2, F=255, G=(1, 2, 4), num_f=400, n_direct=90, shuffle=True, flips=0.3),
    dict(n=3, F=150, G=(1, 1, 2, 2), num_f=256, n_direct=60, density=0.2),
    dict(n=2, F=120, G=(1, 1, 2), num_f=192, n_direct=60, live=True, ta=(2, 4), tb=(4, 8), tc=(4, 8), td=(0, 2)),   # product pairs
    dict(n=3, F=90, G=(1, 1, 1, 2), num_f=128, n_direct=29, live=True, ta=(2, 2), tb=(2, 6), tc=(2, 6), td=(1, 3), shuffle=True),
    dict(n=8, F=100, G=(1, 1, 2, 2, 2, 3, 3, 3, 4), num_f=256, n_direct=180, shuffle=True, identity=False),  # too many bits: per-graph tables
]


@pytest.mark.parametrize("shape", SHARED_SHAPES)
def test_shared_column_table_equals_the_per_graph_tables(hip, shape):
    """When every graph's parity bits fit ONE 16-byte column entry (tsim_program.hip, WR_CCOL) the dense and generic passes of
    k_sample_wide walk a row's set bits once for all graphs; TSIM_AMD_TUNE=wide_compact=0 keeps one table per graph.  Both
    against the oracle, every row class (p_bit up to heavy rows), with the normalisation check."""
    prog = one_wide(77, **shape)
    nf = shape.get("num_f", 320)
    F = shape.get("F", 200)
    orc = OC.OracleProgram(prog)
    B, n = 3000, 5
    fs = [synth.synth_f(B, nf, (1.0 + 3.0 * i) / F, seed=90 + i) for i in range(n)]
    key = prng.key(8)
    subs = _subkeys(key, n)
    packed = ((prog.num_outputs + 7) // 8) % 4 == 0
    got = {}
    prev = os.environ.get("TSIM_AMD_TUNE")
    try:
        for compact in (1, 0):
            os.environ["TSIM_AMD_TUNE"] = f"wide_compact={compact}"
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                hp = hip.HipProgram(prog)
                info = hp.info()
                assert info["wide_fused_kernel"]
                if compact == 0:
                    assert not info["wide_shared_columns"]
                elif len(shape.get("G", (1, 2, 2, 3))) < 9:
                    assert info["wide_shared_columns"], "this shape's graphs fit one entry"
                got[compact] = _steps(hp, prog, fs, key, nf, packed=packed, devs=True)
                hp.close()
    finally:
        if prev is None:
            os.environ.pop("TSIM_AMD_TUNE", None)
        else:
            os.environ["TSIM_AMD_TUNE"] = prev
    for i in range(n):
        want, wdev = orc.sample_program(fs[i], subs[i], return_devs=True)
        for compact in (1, 0):
            np.testing.assert_array_equal(got[compact][0][i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i}, wide_compact={compact}")
            assert np.float32(got[compact][1][i]) == np.float32(wdev[0])


def test_c5_at_scale_equals_the_round2_path_and_the_row_kernel(hip):
    """BASELINE config C5 at 2 x 10^5 shots per batch, 5 batches, shot offsets: the fused kernel, the three-kernel path
    of round 2 (TSIM_AMD_TUNE=wide_fused=0) and the row kernel agree byte for byte; a slice against the oracle."""
    prog, cfg = synth.config_program("C5")
    nf = cfg["num_f"]
    B, n = 200_000, 5
    fs = [synth.synth_f(B, nf, cfg["p_bit"] * (0.5 + 0.5 * i), seed=60 + i) for i in range(n)]
    key = prng.key(8)
    subs = _subkeys(key, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = hip.HipProgram(prog)
        new, _ = _steps(hp, prog, fs, key, nf, packed=True, shot_offset=123)
        again, _ = _steps(hp, prog, fs, key, nf, packed=True, shot_offset=123)  # tables one weight deeper by now: same bits
        os.environ["TSIM_AMD_TUNE"] = "wide_fused=0"
        try:
            hp2 = hip.HipProgram(prog)
        finally:
            os.environ.pop("TSIM_AMD_TUNE", None)
        old, _ = _steps(hp2, prog, fs, key, nf, packed=True, shot_offset=123)
        rows = hip.HipProgram(prog, mode="rows")
        orc = OC.OracleProgram(prog)
        for i in range(n):
            assert np.array_equal(new[i], old[i]) and np.array_equal(new[i], again[i]), f"batch {i}"
            want = rows.sample_batch(fs[i], subs[i], shot_offset=123, bit_packed=True)[0]
            assert np.array_equal(new[i], want), f"batch {i} vs the row kernel"
            w = orc.sample_program(fs[i][:3000], subs[i], shot_offset=123)
            np.testing.assert_array_equal(new[i][:3000], np.packbits(w, axis=1, bitorder="little"))
        for h in (hp, hp2, rows):
            h.close()


def test_ragged_and_tiny_batches(hip):
    """Batches that end inside a 64-row chunk, one-row batches, more batches than a group holds."""
    prog = one_wide(77)
    orc = OC.OracleProgram(prog)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = hip.HipProgram(prog)
        for B, n in ((1, 3), (63, 5), (65, 19), (1000, 2)):
            fs = [synth.synth_f(B, 320, 0.02, seed=B + i) for i in range(n)]
            key = prng.key(B)
            subs = _subkeys(key, n)
            outs, _ = _steps(hp, prog, fs, key, 320, packed=True)
            for i in range(n):
                np.testing.assert_array_equal(outs[i], np.packbits(orc.sample_program(fs[i], subs[i]), axis=1, bitorder="little"),
                                              err_msg=f"B={B} batch {i}")
        hp.close()
