"""The sparse-column kernel for wide components (k_sample4w: more than 64 parameters) against the oracle:
sparse and dense error rows (rows with more than 10 set f bits and the normalisation-check row go through the
row lists to the row kernel), several components, shard offsets, bit-packed output, pipelined launches and a
post-selection row list as input."""

import warnings

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import synth

pytestmark = pytest.mark.gpu


def wide_program(seed=0, physical=True):
    comps = [dict(n=3, F=200, G=[1, 2, 2, 3], density=0.08), dict(n=2, F=90, G=[2, 3, 4], density=0.1)]
    make = synth.physical_program if physical else synth.synth_program
    return make(num_f=320, n_direct=100, components=comps, seed=seed)


@pytest.mark.parametrize("p_bit", [0.0, 0.005, 0.02, 0.06, 0.2])
@pytest.mark.parametrize("physical", [True, False])
def test_wide_sparse_matches_oracle(hip, p_bit, physical):
    prog = wide_program(3, physical)
    f = synth.synth_f(4000, 320, p_bit, seed=int(p_bit * 1000) + 1)
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, (5, 6), return_devs=True, return_overflow=True)
    assert not ov
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = hip.HipProgram(prog)
        assert hp.info()["wide_sparse_kernel"] == physical  # the random program's Dickson forms exceed 32 pairs
        got, gdev = hp.sample_batch(f, (5, 6))
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
        # shards + bit-packed output
        a, _ = hp.sample_batch(f[:1500], (5, 6), bit_packed=True)
        b, _ = hp.sample_batch(f[1500:], (5, 6), shot_offset=1500, bit_packed=True)
        rb = (prog.num_outputs + 7) // 8
        np.testing.assert_array_equal(np.concatenate([a, b])[:, :rb], np.packbits(want, axis=1, bitorder="little"))


def test_wide_sparse_is_selected_and_equals_row_kernel_at_scale(hip):
    prog, cfg = synth.config_program("C5")
    f = synth.synth_f(200_000, cfg["num_f"], cfg["p_bit"], seed=9)
    hp = hip.HipProgram(prog)
    assert hp.info()["wide_sparse_kernel"] and not hp.info()["chunk_table_kernel"]
    a, _ = hp.sample_batch(f, (1, 2), bit_packed=True)
    b, _ = hip.HipProgram(prog, mode="rows").sample_batch(f, (1, 2), bit_packed=True)
    assert np.array_equal(a, b)


def test_wide_sparse_pipelined_and_row_list_input(hip):
    prog = wide_program(5)
    nf, n_out = 320, prog.num_outputs
    wf, wo, rb = (nf + 63) // 64, (n_out + 63) // 64, (n_out + 7) // 8
    hp = hip.HipProgram(prog)
    assert hp.info()["wide_sparse_kernel"]
    B = 30_000
    orc = OC.OracleProgram(prog)
    bufs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(5):
            f = synth.synth_f(B, nf, 0.01 * (i + 1), seed=70 + i)
            pk = np.zeros((B, wf * 8), np.uint8)
            q = np.packbits(f, axis=1, bitorder="little")
            pk[:, : q.shape[1]] = q
            d_f, d_o = hp.malloc(pk.nbytes), hp.malloc(B * rb + 16)
            hp.h2d(d_f, pk)
            hp.sample_batch_device_begin(i, d_f.ptr, B, nf, (i, 4), d_o.ptr, out_bit_packed=True)
            bufs.append((f, d_f, d_o))
        for i in range(5):
            hp.sample_batch_device_end(i)
        hp.synchronize()
        for i, (f, d_f, d_o) in enumerate(bufs):
            got = np.zeros((B, rb), np.uint8)
            hp.d2h(got, d_o)
            np.testing.assert_array_equal(got, np.packbits(orc.sample_program(f, (i, 4)), axis=1, bitorder="little"))
        # post-selection: filter kernel -> survivor list -> sampling on the listed rows only
        f, d_f, _ = bufs[2]
        mask = np.zeros(wo * 64, np.uint8)
        mask[[0, 5]] = 1
        mask_w = np.packbits(mask, bitorder="little").view(np.uint64)
        d_mask, d_rows, d_idx, d_cnt, d_disc = hp.malloc(wo * 8), hp.malloc(B * wo * 8), hp.malloc(B * 4), hp.malloc(4), hp.malloc(B)
        hp.h2d(d_mask, mask_w)
        hp.postselect_device(d_f.ptr, B, nf, d_mask.ptr, 0, d_rows.ptr, d_idx.ptr, d_cnt.ptr, d_disc.ptr)
        hp.sample_rows_device(d_f.ptr, B, nf, (2, 4), d_rows.ptr, d_idx.ptr, d_cnt.ptr)
        hp.synchronize()
        out = np.zeros((B, wo * 8), np.uint8)
        hp.d2h(out, d_rows)
        disc = np.zeros(B, np.uint8)
        hp.d2h(disc, d_disc)
        bits = np.unpackbits(out, axis=1, bitorder="little")[:, :n_out].astype(bool)
        want = orc.sample_program(f, (2, 4))
        keep = disc == 0
        assert keep.any() and (~keep).any()
        np.testing.assert_array_equal(bits[keep], np.asarray(want)[keep])
