"""Low-weight error-pattern tables (tsim_lw.hip.h): the two-pass launch must give exactly the bits
and the normalisation deviation of the full kernel and of the oracle, for every table depth."""

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def _oracle(prog, f, key):
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, key, return_devs=True, return_overflow=True)
    assert not ov
    return want, np.asarray(wdev, np.float32)


def test_tables_selected_for_c2(hip, monkeypatch):
    prog, _ = synth.config_program("C2")
    # (since round 5 a handle STARTS shallow - C2: weight 4 - and reaches weight 5 in the background, test_shallow_start_* below;
    # TSIM_AMD_TUNE=shallow=0 builds the default depth at finalize as rounds 2-4 did)
    assert hip.HipProgram(prog).info()["pattern_max_weight"] == [4]
    monkeypatch.setenv("TSIM_AMD_TUNE", "shallow=0")
    info = hip.HipProgram(prog).info()
    assert info["pattern_tables"] and info["pattern_max_weight"] == [5]
    assert 0 < info["pattern_table_bytes"] <= 256 << 20
    assert not hip.HipProgram(prog, pattern_tables=False).info()["pattern_tables"]
    assert not hip.HipProgram(prog, mode="faithful").info()["pattern_tables"]  # default: auto mode only
    assert hip.HipProgram(prog, mode="faithful", pattern_tables=True).info()["pattern_tables"]
    assert hip.HipProgram(prog, pattern_tables=1).info()["pattern_max_weight"] == [1]


@pytest.mark.parametrize("p_bit", [0.0, 0.01, 0.05, 0.2])
@pytest.mark.parametrize("cap", [0, 1, 2, 3, 4, 5, 6, 7])
def test_c2_every_depth_matches_oracle(hip, p_bit, cap):
    prog, cfg = synth.config_program("C2")
    B = 3000
    f = synth.synth_f(B, cfg["num_f"], p_bit, seed=7 + cap)
    key = prng.key(1234)
    want, wdev = _oracle(prog, f, key)
    hp = hip.HipProgram(prog, pattern_tables=cap)
    assert hp.info()["pattern_max_weight"] == [cap]
    got, gdev = hp.sample_batch(f, key)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), wdev)
    hp.close()  # weight-7 tables are 0.6 GB: do not keep eight handles alive


def test_tables_deepen_on_demand_and_results_do_not_change(hip, monkeypatch):
    """Default depth: 5 at finalize; three launches in a row that leave more than 1 % of their rows to the full
    kernel make the planner build the weight-6/7 tables - in the background.  Launches before, during and after equal the oracle,
    on the serial and on the pipelined API, and sparse launches afterwards still take the short path."""
    prog, cfg = synth.config_program("C2")
    nf, n_out = cfg["num_f"], prog.num_outputs
    monkeypatch.setenv("TSIM_AMD_TUNE", "shallow=0")
    hp = hip.HipProgram(prog)
    monkeypatch.delenv("TSIM_AMD_TUNE")
    assert hp.info()["pattern_max_weight"] == [5]
    B = 20_000
    i = 0
    while i < 8 or (hp.info()["pattern_max_weight"] != [7] and i < 600):
        # (the build runs in the background, one slice per launch plan, the weight-5 tables serving meanwhile: tsim_tables.hip)
        f = synth.synth_f(B, nf, 0.12, seed=40 + i % 16)
        got, gdev = hp.sample_batch(f, (i, 5))
        if i < 8 or i % 16 == 0:
            want, wdev = _oracle(prog, f, (i, 5))
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(np.asarray(gdev, np.float32), wdev)
        i += 1
    assert hp.info()["pattern_max_weight"] == [7], f"no deeper tables after {i} launches"
    for j in range(2):  # the first launches on the new tables
        f = synth.synth_f(B, nf, 0.12, seed=70 + j)
        want, wdev = _oracle(prog, f, (j, 6))
        got, gdev = hp.sample_batch(f, (j, 6))
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), wdev)
    assert hp.info()["pattern_table_bytes"] > 400 << 20
    wf, wo = (nf + 63) // 64, (n_out + 63) // 64
    for p_bit in (0.02, 0.12, 0.02):
        bufs = []
        for i in range(6):
            f = synth.synth_f(B, nf, p_bit, seed=90 + i)
            pk = np.zeros((B, wf * 8), np.uint8)
            q = np.packbits(f, axis=1, bitorder="little")
            pk[:, : q.shape[1]] = q
            d_f, d_o = hp.malloc(pk.nbytes), hp.malloc(B * wo * 8)
            hp.h2d(d_f, pk)
            hp.sample_batch_device_begin(i, d_f.ptr, B, nf, (i, 9), d_o.ptr)
            bufs.append((f, d_f, d_o))
        for i, (f, d_f, d_o) in enumerate(bufs):
            hp.sample_batch_device_end(i)
        hp.synchronize()
        for i, (f, d_f, d_o) in enumerate(bufs):
            out = np.zeros((B, wo * 8), np.uint8)
            hp.d2h(out, d_o)
            got = np.unpackbits(out, axis=1, bitorder="little")[:, :n_out].astype(bool)
            np.testing.assert_array_equal(got, _oracle(prog, f, (i, 9))[0])
            d_f.free()
            d_o.free()
    hp.close()


@pytest.mark.parametrize("launches", [3, 4, 6, 12])
def test_handle_closed_while_deeper_tables_are_being_built(hip, launches):
    """The background build (helper thread allocating, slices on the lanes) must let go whenever the handle goes: closed right
    after the plan asked, mid-allocation, mid-build.  The rows sampled until then are the oracle's."""
    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    B = 20_000
    for rep in range(3):
        hp = hip.HipProgram(prog)
        for i in range(launches):
            f = synth.synth_f(B, nf, 0.12, seed=500 + i)
            got, _ = hp.sample_batch(f, (rep, i))
        want, _ = _oracle(prog, f, (rep, launches - 1))
        np.testing.assert_array_equal(got, want)
        hp.close()


@pytest.mark.parametrize("mode", ["auto", "rows", "faithful"])
def test_tables_with_every_full_kernel(hip, mode):
    prog, cfg = synth.config_program("C2")
    f = synth.synth_f(2000, cfg["num_f"], 0.06, seed=3)
    key = prng.key(99)
    want, wdev = _oracle(prog, f, key)
    on = hip.HipProgram(prog, mode=mode, pattern_tables=True)
    off = hip.HipProgram(prog, mode=mode, pattern_tables=False)
    a, da = on.sample_batch(f, key)
    b, db = off.sample_batch(f, key)
    np.testing.assert_array_equal(a, want)
    np.testing.assert_array_equal(b, want)
    np.testing.assert_array_equal(np.asarray(da, np.float32), wdev)
    np.testing.assert_array_equal(np.asarray(db, np.float32), wdev)


def test_multi_component_and_shards(hip):
    prog, cfg = synth.config_program("C4")
    B = 1500
    f = synth.synth_f(B, cfg["num_f"], 0.03, seed=11)
    key = prng.key(5)
    want, wdev = _oracle(prog, f, key)
    hp = hip.HipProgram(prog)
    assert hp.info()["pattern_tables"]
    got, gdev = hp.sample_batch(f, key)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), wdev)
    # any split of the batch with shot_offset reproduces the unsplit bits (no check row for offset > 0)
    parts = [hp.sample_batch(f[a:b], key, shot_offset=a)[0] for a, b in [(0, 1), (1, 700), (700, 1500)]]
    np.testing.assert_array_equal(np.concatenate(parts), want)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_tables_vs_oracle(hip, seed):
    rng = np.random.default_rng(5000 + seed)
    num_f = int(rng.integers(1, 90))
    comps = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(0, 7))
        F = int(rng.integers(0, min(num_f, 50) + 1))
        comps.append(dict(n=n, F=F, G=[int(rng.integers(0, 6)) for _ in range(n + 1)],
                          ta=(0, 6), tb=(0, 6), tc=(0, 6), td=(0, 3), density=0.3,
                          approx=bool(seed % 3 == 0)))
    prog = synth.synth_program(num_f=num_f, n_direct=int(rng.integers(0, num_f + 1)), components=comps,
                               seed=int(rng.integers(0, 2**31)), shuffle_outputs=True,
                               direct_flip_fraction=0.3, identity_direct=False)
    B = int(rng.choice([1, 64, 65, 300, 1000]))
    f = synth.synth_f(B, num_f, float(rng.choice([0.0, 0.02, 0.08])), seed=seed)
    key = (int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)))
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, key, return_devs=True, return_overflow=True)
    if ov:
        pytest.skip("the reference's int32 arithmetic would wrap on this input")
    for cap in (3, 1):
        hp = hip.HipProgram(prog, pattern_tables=cap)
        assert hp.info()["pattern_tables"]
        got, gdev = hp.sample_batch(f, key)
        np.testing.assert_array_equal(got, want, err_msg=f"cap={cap}")
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))


def test_postselection_list_feeds_pass_one(hip):
    """Survivor list -> pass 1 -> hard list -> full kernel: same bits as the unfiltered launch on the
    surviving rows."""
    from tsim_amd.sampler import CompiledDetectorSampler

    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    probs = [np.array([0.97, 0.03]) for _ in range(nf)]
    et = np.eye(nf, dtype=np.uint8)
    mask = np.zeros(prog.num_detectors, dtype=bool)
    mask[:3] = True
    outs = []
    for pt in (None, False):
        import os

        if pt is False:
            os.environ["TSIM_AMD_PATTERN_TABLES"] = "0"
        try:
            prog2, _ = synth.config_program("C2")  # fresh object: no cached handle
            s = CompiledDetectorSampler(prog2, channel_probs=probs, error_transform=et, seed=3, noise="device")
            outs.append(s.sample(20000, batch_size=8192, postselection_mask=mask, append_observables=True))
        finally:
            os.environ.pop("TSIM_AMD_PATTERN_TABLES", None)
    a, b = outs
    np.testing.assert_array_equal(a, b)
    assert a[:, prog.num_detectors:].any()


def test_pipelined_begin_end_matches_serial(hip):
    """tsim_sample_batch_device_begin/_end: several launches in flight on different slots give the
    bits of the serial call."""
    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    nf, B = cfg["num_f"], 5000
    wf, wo = (nf + 63) // 64, (prog.num_outputs + 63) // 64
    keys = [prng.key(100 + i) for i in range(7)]
    fs = [synth.synth_f(B, nf, 0.03 + 0.02 * (i % 3), seed=40 + i) for i in range(7)]
    want = [hp.sample_batch(f, k, bit_packed=True)[0] for f, k in zip(fs, keys)]
    d_f = [hp.malloc(B * wf * 8) for _ in fs]
    d_o = [hp.malloc(B * wo * 8) for _ in fs]
    for d, f in zip(d_f, fs):
        packed = np.packbits(f, axis=1, bitorder="little")
        packed = np.pad(packed, ((0, 0), (0, wf * 8 - packed.shape[1])))
        hp.h2d(d, packed)
    nslot = hp.PIPELINE_SLOTS
    for i in range(7):
        if i >= nslot:
            hp.sample_batch_device_end((i - nslot) % nslot)
        hp.sample_batch_device_begin(i % nslot, d_f[i].ptr, B, nf, keys[i], d_o[i].ptr)
    for s in range(nslot):
        hp.sample_batch_device_end(s)
    hp.synchronize()
    for i in range(7):
        got = np.zeros((B, wo * 8), np.uint8)
        hp.d2h(got, d_o[i])
        np.testing.assert_array_equal(got, want[i], err_msg=f"launch {i}")


@pytest.mark.parametrize("p_bit", [0.01, 0.4])
def test_launch_plan_adapts_without_changing_bits(hip, p_bit):
    """Repeated launches on one handle: the plan (two-pass / direct / overflow launch) follows the
    feedback of earlier launches; every launch still equals the oracle."""
    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    for i in range(20):
        B = 2000 if i % 5 else 3000
        f = synth.synth_f(B, cfg["num_f"], p_bit, seed=900 + i)
        key = prng.key(i)
        got, gdev = hp.sample_batch(f, key)
        if i % 6 == 0 or i == 19:
            want, wdev = _oracle(prog, f, key)
            np.testing.assert_array_equal(got, want, err_msg=f"launch {i}")
            np.testing.assert_array_equal(np.asarray(gdev, np.float32), wdev)
        else:
            other = hip.HipProgram(prog, pattern_tables=False).sample_batch(f, key)[0] if i == 7 else None
            if other is not None:
                np.testing.assert_array_equal(got, other)


def test_stage_profiling_reports_every_kernel(hip):
    """tsim_profile_read_stages: the pattern pass and the hard-row kernel both show up (a lost event
    would silently turn the roofline figure of bench.py into the step time)."""
    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    f = synth.synth_f(50000, cfg["num_f"], 0.02, seed=1)
    hp.sample_batch(f, prng.key(1))
    hp.profile_enable(1)
    hp.profile_read(reset=True)
    for i in range(3):
        hp.sample_batch(f, prng.key(2 + i))
    st = hp.profile_read_stages()
    ms, n = hp.profile_read(reset=True)
    hp.profile_enable(False)
    assert n == 3 and st["pattern_pass"] > 0 and st["hard_rows"] > 0 and ms >= st["pattern_pass"] + st["hard_rows"] - 1e-3


@pytest.mark.parametrize("tables", [True, False])
def test_compact_output_written_by_the_kernels(hip, tables):
    """tsim_pipeline_set_compact_output: the sampling kernels (pattern pass, hard-row kernel, full kernel)
    write the bit_packed rows themselves - equal to packbits of the normal output."""
    import ctypes as C

    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog, pattern_tables=tables)
    nf, B = cfg["num_f"], 30011
    wf, wo, rb = (nf + 63) // 64, (prog.num_outputs + 63) // 64, (prog.num_outputs + 7) // 8
    for p_bit in (0.02, 0.3):
        f = synth.synth_f(B, nf, p_bit, seed=17)
        key = prng.key(8)
        want = hp.sample_batch(f, key)[0]
        fp = np.packbits(f, axis=1, bitorder="little")
        fp = np.ascontiguousarray(np.pad(fp, ((0, 0), (0, wf * 8 - fp.shape[1]))))
        d_f, d_o, d_c = hp.malloc(B * wf * 8), hp.malloc(B * wo * 8), hp.malloc(B * rb + 16)
        hp.h2d(d_f, fp)
        assert hp._lib.tsim_pipeline_set_compact_output(hp._h, 1, C.c_void_p(d_c.ptr)) == 0
        hp.sample_batch_device_begin(1, d_f.ptr, B, nf, key, d_o.ptr)
        hp.sample_batch_device_end(1)
        hp.synchronize()
        got = np.zeros((B, rb), np.uint8)
        hp.d2h(got, d_c)
        np.testing.assert_array_equal(got, np.packbits(want, axis=1, bitorder="little"))


def _pipelined_run(hp, prog, cfg, n_launch, B, p_bits, order_end_first, nslot, compact=False, with_check=True):
    """n_launch pipelined launches on `nslot` slots; returns (outputs, compact outputs, norm devs)."""
    import ctypes as C

    nf = cfg["num_f"]
    wf, wo, rb = (nf + 63) // 64, (prog.num_outputs + 63) // 64, (prog.num_outputs + 7) // 8
    n_comp = max(1, len(prog.components))
    fs = [synth.synth_f(B, nf, p_bits[i % len(p_bits)], seed=300 + i) for i in range(n_launch)]
    keys = [prng.key(500 + i) for i in range(n_launch)]
    d_f = [hp.malloc(B * wf * 8) for _ in fs]
    d_o = [hp.malloc(B * wo * 8) for _ in fs]
    d_c = [hp.malloc(B * rb + 16) for _ in fs]
    d_dev = hp.malloc(n_launch * n_comp * 4)
    for d, f in zip(d_f, fs):
        packed = np.packbits(f, axis=1, bitorder="little")
        hp.h2d(d, np.ascontiguousarray(np.pad(packed, ((0, 0), (0, wf * 8 - packed.shape[1])))))
    for i in range(n_launch):
        slot = i % nslot
        if order_end_first and i >= nslot:
            hp.sample_batch_device_end(slot)
        if compact:
            assert hp._lib.tsim_pipeline_set_compact_output(hp._h, slot, C.c_void_p(d_c[i].ptr)) == 0
        hp.sample_batch_device_begin(slot, d_f[i].ptr, B, nf, keys[i], d_o[i].ptr,
                                     shot_offset=0 if with_check else 64 * i + 64,
                                     d_norm_dev=d_dev.ptr + i * n_comp * 4)
    for s in range(nslot):
        hp.sample_batch_device_end(s)
    hp.synchronize()
    outs, comps = [], []
    for i in range(n_launch):
        got = np.zeros((B, wo * 8), np.uint8)
        hp.d2h(got, d_o[i])
        outs.append(got)
        if compact:
            c = np.zeros((B, rb), np.uint8)
            hp.d2h(c, d_c[i])
            comps.append(c)
    devs = np.zeros(n_launch * n_comp, np.float32)
    hp.d2h(devs, d_dev)
    return fs, keys, outs, comps, devs.reshape(n_launch, n_comp)


@pytest.mark.parametrize("group", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("nslot", [1, 3, 8])
def test_deferred_hard_row_batches(hip, monkeypatch, group, nslot):
    """The hard rows of several pipelined launches served by ONE k_sample4h_multi grid (deferred second
    pass): every batch size, slot count and join order gives the bits of the serial full kernel,
    including the normalisation check of each launch and the bit_packed second output."""
    monkeypatch.setenv("TSIM_AMD_TUNE", f"defer_group={group}")
    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    ref = hip.HipProgram(prog, pattern_tables=False)
    # warm the feedback so that the plan switches to deferred batches (first launches run their own pass)
    for order_end_first in (True, False):
        fs, keys, outs, comps, devs = _pipelined_run(hp, prog, cfg, 13, 20011, [0.02, 0.03, 0.01], order_end_first,
                                                     nslot, compact=True)
        for i, (f, k) in enumerate(zip(fs, keys)):
            want, wdev = ref.sample_batch(f, k, bit_packed=True)
            np.testing.assert_array_equal(outs[i], want, err_msg=f"launch {i}")
            np.testing.assert_array_equal(comps[i], want[:, : comps[i].shape[1]], err_msg=f"compact {i}")
            np.testing.assert_array_equal(devs[i], np.asarray(wdev, np.float32), err_msg=f"norm dev {i}")


def test_deferred_batches_off_and_on_agree(hip, monkeypatch):
    prog, cfg = synth.config_program("C3")
    runs = []
    for defer in ("1", "0"):
        monkeypatch.setenv("TSIM_AMD_TUNE", f"defer_hard={defer}")
        hp = hip.HipProgram(prog)
        runs.append(_pipelined_run(hp, prog, cfg, 11, 9001, [0.02, 0.002], True, 5, with_check=False))
    for a, b in zip(runs[0][2], runs[1][2]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("p_bit,lists_expected", [(0.002, "few"), (0.06, "many")])
def test_hard_row_list_count_follows_the_load(hip, p_bit, lists_expected):
    """Few hard rows -> few, well filled lists; many -> all 64.  Same bits either way (serial API)."""
    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    ref = hip.HipProgram(prog, pattern_tables=False)
    for i in range(6):
        f = synth.synth_f(40000, cfg["num_f"], p_bit, seed=70 + i)
        got = hp.sample_batch(f, prng.key(i))[0]
        np.testing.assert_array_equal(got, ref.sample_batch(f, prng.key(i))[0])


def test_join_on_the_batch_lane_and_wait_stream(hip):
    """A consumer may join its slots on lane 2 (where the deferred batches run) instead of the handle's
    stream, and order every lane after a stream with one call; results as ever."""
    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    ref = hip.HipProgram(prog, pattern_tables=False)
    nf, B = cfg["num_f"], 12007
    wf, wo = (nf + 63) // 64, (prog.num_outputs + 63) // 64
    lane2 = hp.pipeline_lane_stream(2)
    assert lane2 != 0 and lane2 != hp.stream_ptr() and hp.pipeline_lane_stream(0) == hp.stream_ptr()
    n, nslot = 21, 6
    fs = [synth.synth_f(B, nf, 0.02, seed=800 + i) for i in range(n)]
    keys = [prng.key(900 + i) for i in range(n)]
    d_f = [hp.malloc(B * wf * 8) for _ in range(nslot)]
    d_o = [hp.malloc(B * wo * 8) for _ in range(n)]
    for i in range(n):
        slot = i % nslot
        if i >= nslot:
            hp.sample_batch_device_end(slot, lane2)
        # the slot's f buffer is rewritten on the handle's stream: every lane must see it
        packed = np.packbits(fs[i], axis=1, bitorder="little")
        if i >= nslot:
            hp.synchronize()  # the slot's previous launch no longer reads the buffer the blocking copy rewrites
        hp.h2d(d_f[slot], np.ascontiguousarray(np.pad(packed, ((0, 0), (0, wf * 8 - packed.shape[1])))))
        hp.pipeline_wait_stream()
        hp.sample_batch_device_begin(slot, d_f[slot].ptr, B, nf, keys[i], d_o[i].ptr, inputs_ready=True)
    for s in range(nslot):
        hp.sample_batch_device_end(s, lane2)
    hp.synchronize()
    for i in range(n):
        got = np.zeros((B, wo * 8), np.uint8)
        hp.d2h(got, d_o[i])
        np.testing.assert_array_equal(got, ref.sample_batch(fs[i], keys[i], bit_packed=True)[0], err_msg=f"launch {i}")


def test_compact_series_one_slice_per_launch(hip):
    """tsim_pipeline_set_compact_series: the next `count` pipelined launches write their bit_packed rows
    into consecutive slices of one buffer (what bench.py gathers), later launches do not."""
    import ctypes as C

    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    nf, B, n = cfg["num_f"], 7001, 6
    wf, wo, rb = (nf + 63) // 64, (prog.num_outputs + 63) // 64, (prog.num_outputs + 7) // 8
    fs = [synth.synth_f(B, nf, 0.02, seed=60 + i) for i in range(n)]
    keys = [prng.key(70 + i) for i in range(n)]
    d_f = [hp.malloc(B * wf * 8) for _ in range(n)]
    d_o = [hp.malloc(B * wo * 8) for _ in range(n)]
    d_c = hp.malloc(n * B * rb + 16)
    hp.h2d(d_c, np.full(n * B * rb + 16, 0xEE, np.uint8))
    for d, f in zip(d_f, fs):
        packed = np.packbits(f, axis=1, bitorder="little")
        hp.h2d(d, np.ascontiguousarray(np.pad(packed, ((0, 0), (0, wf * 8 - packed.shape[1])))))
    assert hp._lib.tsim_pipeline_set_compact_series(hp._h, C.c_void_p(d_c.ptr), B * rb, n - 1) == 0
    for i in range(n):
        hp.sample_batch_device_begin(i % 4, d_f[i].ptr, B, nf, keys[i], d_o[i].ptr)
    for s in range(4):
        hp.sample_batch_device_end(s)
    hp.synchronize()
    got = np.zeros(n * B * rb + 16, np.uint8)
    hp.d2h(got, d_c)
    for i in range(n - 1):
        want = hp.sample_batch(fs[i], keys[i])[0]
        np.testing.assert_array_equal(got[i * B * rb:(i + 1) * B * rb].reshape(B, rb), np.packbits(want, axis=1, bitorder="little"))
    assert (got[(n - 1) * B * rb:] == 0xEE).all()  # the series had n - 1 entries


def test_destroy_with_parked_hard_rows_and_mixed_serial_calls(hip):
    """Handles dropped while launches are parked (never joined), and the serial API used between
    pipelined launches of the same handle."""
    prog, cfg = synth.config_program("C2")
    ref = hip.HipProgram(prog, pattern_tables=False)
    nf, B = cfg["num_f"], 9001
    wf, wo = (nf + 63) // 64, (prog.num_outputs + 63) // 64
    f = synth.synth_f(B, nf, 0.02, seed=5)
    packed = np.packbits(f, axis=1, bitorder="little")
    packed = np.ascontiguousarray(np.pad(packed, ((0, 0), (0, wf * 8 - packed.shape[1]))))
    for round_ in range(3):
        hp = hip.HipProgram(prog)
        d_f, outs = hp.malloc(B * wf * 8), [hp.malloc(B * wo * 8) for _ in range(7)]
        hp.h2d(d_f, packed)
        for i in range(6):  # warm the plan, join properly
            hp.sample_batch_device_begin(i % 3, d_f.ptr, B, nf, prng.key(i), outs[i].ptr)
        for s in range(3):
            hp.sample_batch_device_end(s)
        serial = hp.sample_batch(f, prng.key(99))[0]           # serial call in between
        np.testing.assert_array_equal(serial, ref.sample_batch(f, prng.key(99))[0])
        for i in range(5):  # parked: no end, no synchronize
            hp.sample_batch_device_begin(i, d_f.ptr, B, nf, prng.key(10 + i), outs[i].ptr)
        if round_ == 1:
            hp.synchronize()
            got = np.zeros((B, wo * 8), np.uint8)
            hp.d2h(got, outs[4])
            np.testing.assert_array_equal(got, ref.sample_batch(f, prng.key(14), bit_packed=True)[0])
        del hp, d_f, outs


def test_begin_split_is_split_then_begin(hip):
    """tsim_sample_batch_device_begin_split: key, subkey = split(key) inside the call (sampler.py:399)."""
    import ctypes as C

    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    nf, B = cfg["num_f"], 6007
    wf, wo = (nf + 63) // 64, (prog.num_outputs + 63) // 64
    f = synth.synth_f(B, nf, 0.02, seed=3)
    packed = np.packbits(f, axis=1, bitorder="little")
    d_f, d_o = hp.malloc(B * wf * 8), [hp.malloc(B * wo * 8) for _ in range(3)]
    hp.h2d(d_f, np.ascontiguousarray(np.pad(packed, ((0, 0), (0, wf * 8 - packed.shape[1])))))
    key = prng.key(2024)
    state = (C.c_uint32 * 2)(key[0], key[1])
    for i in range(3):
        assert hp._lib.tsim_sample_batch_device_begin_split(hp._h, i, d_f.ptr, B, nf, state, 0, d_o[i].ptr, None, None, 0) == 0
    for i in range(3):
        hp.sample_batch_device_end(i)
    hp.synchronize()
    for i in range(3):
        key, sub = prng.split(key)
        got = np.zeros((B, wo * 8), np.uint8)
        hp.d2h(got, d_o[i])
        np.testing.assert_array_equal(got, hp.sample_batch(f, sub, bit_packed=True)[0], err_msg=f"batch {i}")
    assert (state[0], state[1]) == (key[0], key[1])


@pytest.mark.parametrize("name", ["C2", "C5"])
def test_bit_packed_only_output_of_pipelined_launches(hip, name):
    """TSIM_PIPE_OUT_BIT_PACKED: the kernels write ceil(n_out/8)-byte rows instead of the padded words - equal to
    np.packbits of the ordinary result, for tabulated rows, hard rows (own pass and deferred batches) and the full kernel."""
    import ctypes as C

    prog, cfg = synth.config_program(name)
    hp = hip.HipProgram(prog)
    nf, n_out = cfg["num_f"], prog.num_outputs
    wf, rb = (nf + 63) // 64, (n_out + 7) // 8
    B, n = 50_000, 12
    d_f, d_c = [], []
    want = []
    key = [(i + 1, 7 * i + 3) for i in range(n)]
    for i in range(n):
        f = synth.synth_f(B, nf, cfg["p_bit"] * (1 + i % 3), seed=300 + i)
        packed = np.zeros((B, wf * 8), np.uint8)
        pk = np.packbits(f, axis=1, bitorder="little")
        packed[:, : pk.shape[1]] = pk
        buf = hp.malloc(packed.nbytes)
        hp.h2d(buf, packed)
        d_f.append(buf)
        d_c.append(hp.malloc(B * rb + 16))
        ref, _ = hp.sample_batch(f, key[i])
        want.append(np.packbits(ref, axis=1, bitorder="little"))
    for rounds in range(2):  # second round: the deferred plan is active (feedback from the first)
        for i in range(n):
            rc = hp._lib.tsim_sample_batch_device_begin(hp._h, i % 8, d_f[i].ptr, B, nf, key[i][0], key[i][1], 0, d_c[i].ptr, None, None, 2)
            assert rc == 0
            if i % 8 == 7:
                for s in range(8):
                    hp.sample_batch_device_end(s)
        for s in range(8):
            hp.sample_batch_device_end(s)
        hp.synchronize()
        for i in range(n):
            got = np.zeros((B, rb), np.uint8)
            hp.d2h(got, d_c[i])
            np.testing.assert_array_equal(got, want[i])


@pytest.mark.parametrize("name,start", [("C2", [4]), ("C4", [3, 3, 3]), ("C3", [3])])
def test_shallow_start_reaches_the_default_depth_and_bits_do_not_change(hip, name, start):
    """finalize builds only the tables that cost about half a millisecond (the BASELINE jobs are 10^5-10^6
    shots; C4's weight-5 tables are 195 MB and 35 ms), the default depth (5) follows in the background, one slice per launch
    plan.  Every batch - before, during and after the swap - equals the oracle, through the fused steps API and the serial one."""
    from test_gpu_steps import _run_steps, _subkeys

    prog, cfg = synth.config_program(name)
    nf = cfg["num_f"]
    hp = hip.HipProgram(prog)
    assert hp.info()["pattern_max_weight"] == start
    op = OC.OracleProgram(prog)
    B, n = 2000, 3
    key = prng.key(31)
    rounds = 0
    seen = {tuple(start)}
    while rounds < 400:
        fs = [synth.synth_f(B, nf, cfg["p_bit"] * (1 + (i % 2)), seed=7 * rounds + i) for i in range(n)]
        outs, key_after = _run_steps(hp, prog, fs, key, nf, packed=True)
        _, subs = _subkeys(key, n)
        if rounds < 6 or rounds % 16 == 0 or tuple(hp.info()["pattern_max_weight"]) not in seen:
            for i in range(n):
                np.testing.assert_array_equal(outs[i], np.packbits(op.sample_program(fs[i], subs[i]), axis=1, bitorder="little"), err_msg=f"round {rounds} batch {i}")
        seen.add(tuple(hp.info()["pattern_max_weight"]))
        key = key_after
        rounds += 1
        if hp.info()["pattern_max_weight"] == [5] * len(start) and rounds > 8:
            break
    assert hp.info()["pattern_max_weight"] == [5] * len(start), f"still {hp.info()['pattern_max_weight']} after {rounds} calls"
    f = synth.synth_f(3000, nf, cfg["p_bit"], seed=99)
    got, gdev = hp.sample_batch(f, (4, 4))
    want, wdev = op.sample_program(f, (4, 4), return_devs=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
    hp.close()


def test_a_handle_destroyed_during_the_background_build(hip):
    """close() right after creation, and after one call: the helper thread and the slices in flight are drained."""
    prog, cfg = synth.config_program("C4")
    for calls in (0, 1, 3):
        hp = hip.HipProgram(prog)
        for i in range(calls):
            hp.sample_batch(synth.synth_f(500, cfg["num_f"], 0.02, seed=i), (i, 1))
        hp.close()


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "F60", "F70", "n9", "6narrow", "n13", "n16"])
def test_chunk_table_builder_equals_row_builder(hip, name, monkeypatch):
    """The finalize build of the pattern tables runs on the LDS chunk tables where the program has them (csrc/tsim_build4.hip);
    ``TSIM_AMD_TABLE_BUILD=rows`` keeps the row formulation's builder.  Same thresholds: the same bytes for batches that the
    tables serve (and the oracle's), at a pinned depth so that no background build interferes.  n13 / n16: prefix-tree tables,
    whose node pass takes the same route (k_trie_nodes4)."""
    from oracle import oracle_c as OC
    from tsim_amd import synth

    if name in ("C2", "C3", "C4"):
        prog, cfg = synth.config_program(name)
    else:
        prog, cfg = synth.shape_class_program(name)
    nf = cfg["num_f"]
    B = 20000
    f = synth.synth_f(B, nf, cfg["p_bit"], seed=77)
    outs = []
    for build in ("rows", ""):
        if build:
            monkeypatch.setenv("TSIM_AMD_TABLE_BUILD", build)
        else:
            monkeypatch.delenv("TSIM_AMD_TABLE_BUILD", raising=False)
        hp = hip.HipProgram(prog, pattern_tables=3)
        hp.path_counts(reset=True)
        out, dev = hp.sample_batch(f, (5, 9))
        outs.append((np.asarray(out), np.asarray(dev), hp.path_counts()))
        hp.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    want = OC.OracleProgram(prog).sample_program(f, (5, 9))
    np.testing.assert_array_equal(outs[1][0], want)
