"""Programs with SEVERAL wide-path components (more than 64 parameters somewhere, or a narrow component next to a wide
one): one k_sample_wide pass per component since round 5 (tsim_wide.hip.h, WR_MERGE) - the first pass writes whole rows,
the later ones OR their component's bits into the rows in place.  The reference samples components one after the other
into one output array (src/tsim/sampler.py:146-163); every case here against the oracle, bit for bit, with the
per-component normalisation deviations, for padded and bit_packed rows of any byte size and alignment, all three row
classes of the kernel (tabulated / missed / heavy), the serial and the several-batches API, and against the round-2
path (`wide_passes=1`)."""

import ctypes as C
import os
import warnings

import numpy as np
import pytest

from oracle import oracle_c as OC
from test_gpu_steps import _run_steps, _subkeys
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def _w(n, F, G=None, density=0.08):
    return dict(n=n, F=F, G=list(G or [1, 2, 2, 3, 3, 4, 4, 5, 5][: n + 1]), density=density)


SHAPES = {
    # two wide components, 123 outputs (16-byte bit_packed rows)
    "2wide": dict(num_f=320, n_direct=118, components=[_w(2, 100), _w(3, 150)]),
    # a narrow component next to a wide one, outputs shuffled over the words, flips, 13-byte rows
    "narrow+wide": dict(num_f=320, n_direct=96, components=[_w(2, 20, [2, 4, 8]), _w(3, 200)], shuffle_outputs=True, direct_flip_fraction=0.3),
    # three components, direct table not the identity, 9-byte rows
    "3mixed": dict(num_f=256, n_direct=61, components=[_w(1, 12, [1, 2]), _w(4, 90), _w(2, 130)], shuffle_outputs=True, identity_direct=False,
                   direct_flip_fraction=0.2),
    # four components and no direct outputs at all: the first pass writes zeros + its component, 2-byte rows
    "4comps": dict(num_f=192, n_direct=0, components=[_w(1, 70), _w(2, 80), _w(3, 66), _w(5, 75)], shuffle_outputs=True),
    # f rows of 1500 bits (24 words; the round-2 kernels read 16 mask words: f indices below 512), direct outputs from anywhere in the row
    "bigrows": dict(num_f=1500, n_direct=100, components=[_w(3, 180), _w(2, 40)], shuffle_outputs=True, identity_direct=False, direct_flip_fraction=0.1),
    # more than 255 selected bits: positions in 16 bits (k_sample_wide<.., P16>; positions were bytes until round 5: the row kernel without tables)
    "F300": dict(num_f=320, n_direct=20, components=[_w(3, 300)]),
    "F400+narrow": dict(num_f=512, n_direct=70, components=[_w(2, 30, [2, 4, 8]), _w(4, 400, [1, 1, 1, 2, 2])], shuffle_outputs=True, direct_flip_fraction=0.2),
    "F511": dict(num_f=640, n_direct=10, components=[_w(2, 511)], identity_direct=False),
    # eight outputs per component, 200 outputs: four output words
    "wide8": dict(num_f=448, n_direct=184, components=[_w(8, 100), _w(8, 120)], shuffle_outputs=True),
}


def _program(name):
    kw = dict(SHAPES[name])
    return synth.physical_program(seed=77, **kw), kw["num_f"]


def _handle(hip, prog, tune=None, **kw):
    old = os.environ.get("TSIM_AMD_TUNE")
    if tune is not None:
        os.environ["TSIM_AMD_TUNE"] = tune
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return hip.HipProgram(prog, **kw)
    finally:
        if tune is not None:
            if old is None:
                os.environ.pop("TSIM_AMD_TUNE", None)
            else:
                os.environ["TSIM_AMD_TUNE"] = old


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("packed", [True, False])
def test_steps_api_equals_oracle(hip, name, packed):
    prog, nf = _program(name)
    orc = OC.OracleProgram(prog)
    B, n = 2300, 10
    fmax = max(len(c.f_selection) for c in prog.components)
    # mean weight of the widest component 2 .. 8: tabulated rows, missed rows (dense passes) and, at the top, heavy rows
    fs = [synth.synth_f(B, nf, (2.0 + 2.0 * (i % 4)) / fmax, seed=300 + i) for i in range(n)]
    key = prng.key(17)
    hp = _handle(hip, prog)
    hp.path_counts(reset=True)
    devs = []
    outs, _ = _run_steps(hp, prog, fs, key, nf, packed=packed, calls=[3, 7], devs=devs)
    paths = hp.path_counts()
    assert paths.get("wide", 0) == 2 * len(prog.components), paths  # two groups, one pass per component each - nothing else
    assert set(paths) == {"wide"}, paths
    _, subs = _subkeys(key, n)
    for i in range(n):
        want, wdev = orc.sample_program(fs[i], subs[i], return_devs=True)
        np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"{name} batch {i}")
        np.testing.assert_array_equal(devs[i][: len(prog.components)], np.asarray(wdev, np.float32), err_msg=f"{name} batch {i}: deviations")
    hp.close()


@pytest.mark.parametrize("name", ["narrow+wide", "3mixed", "4comps", "F300", "F400+narrow"])
@pytest.mark.parametrize("p_scale", [0.0, 1.0, 6.0])
def test_serial_api_shards_and_dense_noise(hip, name, p_scale):
    """tsim_sample_batch: hits only (no set bit at all), the nominal level, and mostly heavy rows (generic passes that merge);
    two shards with a shot offset (no normalisation check in the second) equal one batch."""
    prog, nf = _program(name)
    orc = OC.OracleProgram(prog)
    fmax = max(len(c.f_selection) for c in prog.components)
    f = synth.synth_f(2600, nf, p_scale * 4.0 / fmax, seed=int(p_scale) + 5)
    want, wdev = orc.sample_program(f, (4, 9), return_devs=True)
    hp = _handle(hip, prog)
    got, gdev = hp.sample_batch(f, (4, 9))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
    a, _ = hp.sample_batch(f[:1001], (4, 9), bit_packed=True)
    b, _ = hp.sample_batch(f[1001:], (4, 9), shot_offset=1001, bit_packed=True)
    rb = (prog.num_outputs + 7) // 8  # (sample_batch returns the padded words of a row as bytes)
    np.testing.assert_array_equal(np.concatenate([a, b])[:, :rb], np.packbits(want, axis=1, bitorder="little"))
    assert not np.concatenate([a, b])[:, rb:].any()
    hp.close()


@pytest.mark.parametrize("name", ["2wide", "3mixed"])
def test_passes_equal_the_round2_path(hip, name):
    """The same batches through the per-component passes and through the round-2 kernels (`wide_passes=1`: table pass,
    sparse-column kernel, row kernel): identical rows, and the second handle really took the other path."""
    prog, nf = _program(name)
    fmax = max(len(c.f_selection) for c in prog.components)
    fs = [synth.synth_f(4000, nf, (3.0 + i) / fmax, seed=90 + i) for i in range(4)]
    key = prng.key(3)
    res = []
    for tune in (None, "wide_passes=1"):
        hp = _handle(hip, prog, tune=tune)
        hp.path_counts(reset=True)
        outs, _ = _run_steps(hp, prog, fs, key, nf, packed=True)
        res.append((outs, hp.path_counts()))
        hp.close()
    assert "wide" in res[0][1] and "wide" not in res[1][1], (res[0][1], res[1][1])
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(a, b)


def test_f_rows_of_2000_bits_and_beyond(hip):
    """2000 f bits with 250 selected ones scattered over 63 of the 63 words: still k_sample_wide (its word list holds 64).
    2500 bits: past every wide kernel's reach - the row kernel serves the program, and the round-2 wide kernels (masks up to
    bit 511) must not be launched.  Same bits as the oracle either way."""
    for num_f, wide in ((2000, True), (2500, False)):
        prog = synth.physical_program(seed=5, num_f=num_f, n_direct=40, components=[_w(3, 250)], identity_direct=False)
        orc = OC.OracleProgram(prog)
        f = synth.synth_f(1200, num_f, 4.0 / 250, seed=2)
        hp = _handle(hip, prog)
        hp.path_counts(reset=True)
        for i in range(3):
            want, wdev = orc.sample_program(f, (i, 1), return_devs=True)
            got, gdev = hp.sample_batch(f, (i, 1))
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
        paths = hp.path_counts()
        assert not ({"sample4w", "lw_lds_wide"} & set(paths)), paths
        assert ("wide" in paths) == wide, (num_f, paths)
        hp.close()


def test_misaligned_bit_packed_buffers(hip):
    """bit_packed rows of 13 bytes written into a buffer that starts at an odd address: every byte of every row right, and
    not a byte outside the rows touched."""
    prog, nf = _program("narrow+wide")
    orc = OC.OracleProgram(prog)
    B, rb, wf = 1777, (prog.num_outputs + 7) // 8, (nf + 63) // 64
    assert rb % 4 != 0
    fmax = max(len(c.f_selection) for c in prog.components)
    f = synth.synth_f(B, nf, 5.0 / fmax, seed=8)
    key = prng.key(29)
    _, subs = _subkeys(key, 1)
    want = np.packbits(orc.sample_program(f, subs[0]), axis=1, bitorder="little")
    hp = _handle(hip, prog)
    d_f = hp.malloc(B * wf * 8)
    p = np.packbits(f, axis=1, bitorder="little")
    hp.h2d(d_f, np.ascontiguousarray(np.pad(p, ((0, 0), (0, wf * 8 - p.shape[1])))))
    for shift in (1, 2, 3, 7):
        d_o = hp.malloc(B * rb + 64)
        guard = np.full(B * rb + 64, 0xA5, np.uint8)
        hp.h2d(d_o, guard)
        ks = (C.c_uint32 * 2)(key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF)
        hp.sample_steps_device([d_f.ptr], B, nf, ks, [d_o.ptr + shift], out_bit_packed=True)
        hp.synchronize()
        raw = np.zeros(B * rb + 64, np.uint8)
        hp.d2h(raw, d_o)
        np.testing.assert_array_equal(raw[shift: shift + B * rb].reshape(B, rb), want, err_msg=f"shift {shift}")
        assert (raw[:shift] == 0xA5).all() and (raw[shift + B * rb:] == 0xA5).all(), f"shift {shift}: bytes outside the rows were written"
        d_o.free()
    d_f.free()
    hp.close()


@pytest.mark.parametrize("name", ["narrow+wide", "4comps", "F300"])
def test_tiny_batches_and_large_shot_offsets(hip, name):
    """Batches of 1, 63, 64, 65 rows (a chunk is 64 rows; the passes of the later components must touch nothing beyond the
    batch) and shards whose shot range lies beyond 2^32 or crosses it (k_sample_wide counts the low word: the launcher
    sends a crossing range to the other kernels) - the counter of the draws is the global shot index, sampler.py:74-75."""
    prog, nf = _program(name)
    orc = OC.OracleProgram(prog)
    fmax = max(len(c.f_selection) for c in prog.components)
    hp = _handle(hip, prog)
    for B, off in ((1, 0), (63, 0), (64, 0), (65, 0), (130, (1 << 32) + 5), (200, (1 << 32) - 100), (97, (1 << 33) - 97)):
        f = synth.synth_f(B, nf, 5.0 / fmax, seed=B + 7)
        want = orc.sample_program(f, (B, 3), shot_offset=off)
        got, _ = hp.sample_batch(f, (B, 3), shot_offset=off)
        np.testing.assert_array_equal(got, want, err_msg=f"{name} B {B} shot_offset {off}")
        rb = (prog.num_outputs + 7) // 8
        pk, _ = hp.sample_batch(f, (B, 3), shot_offset=off, bit_packed=True)
        np.testing.assert_array_equal(pk[:, :rb], np.packbits(want, axis=1, bitorder="little"), err_msg=f"{name} B {B} shot_offset {off} (bit_packed)")
    hp.close()
