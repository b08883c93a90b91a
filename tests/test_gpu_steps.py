"""tsim_sample_steps_device: several consecutive batches of the reference's batch loop (sampler.py:340-420) in one
call - fused first passes (k_sample_lw_multi) where the register form applies - must give, batch for batch, the
bits of the oracle with the split key chain of sampler.py:399 (and therefore of the one-batch-per-call API)."""

import ctypes as C

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def _packed(f, wf):
    p = np.packbits(f, axis=1, bitorder="little")
    return np.ascontiguousarray(np.pad(p, ((0, 0), (0, wf * 8 - p.shape[1]))))


def _subkeys(key, n):
    """key, sub = split(key) once per batch (sampler.py:399)."""
    subs = []
    for _ in range(n):
        key, sub = prng.split(key)
        subs.append(sub)
    return key, subs


def _run_steps(hp, prog, fs, key, nf, *, packed, shot_offset=0, calls=None, devs=None):
    """fs through sample_steps_device (optionally split into several calls of the given sizes); returns the rows
    (devs: a list that receives every batch's normalisation deviations, one float32 array per batch)."""
    B = len(fs[0])
    wf, wo, rb = (nf + 63) // 64, (prog.num_outputs + 63) // 64, (prog.num_outputs + 7) // 8
    n_comp = max(1, len(prog.components))
    d_f = [hp.malloc(B * wf * 8) for _ in fs]
    d_o = [hp.malloc(max(B * wo * 8, 16)) for _ in fs]
    d_d = [hp.malloc(4 * n_comp + 16) for _ in fs] if devs is not None else []
    for d, f in zip(d_f, fs):
        hp.h2d(d, _packed(f, wf))
    ks = (C.c_uint32 * 2)(key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF)
    i = 0
    for n in (calls or [len(fs)]):
        hp.sample_steps_device([d.ptr for d in d_f[i:i + n]], B, nf, ks, [d.ptr for d in d_o[i:i + n]],
                               shot_offset=shot_offset, out_bit_packed=packed,
                               d_norm_dev=[d.ptr for d in d_d[i:i + n]] if devs is not None else None)
        i += n
    assert i == len(fs)
    hp.synchronize()
    for d in d_d:
        v = np.zeros(n_comp, np.float32)
        hp.d2h(v, d)
        devs.append(v)
        d.free()
    outs = []
    for d in d_o:
        if packed:
            got = np.zeros((B, rb), np.uint8)
            hp.d2h(got, d)
        else:
            raw = np.zeros((B, wo * 8), np.uint8)
            hp.d2h(raw, d)
            got = np.packbits(np.unpackbits(raw, axis=1, bitorder="little")[:, : prog.num_outputs], axis=1, bitorder="little")
        outs.append(got)
    for d in d_f + d_o:
        d.free()
    return outs, (int(ks[0]), int(ks[1]))


@pytest.mark.parametrize("name,B,n", [("C2", 3000, 11), ("C3", 2500, 9), ("C4", 1500, 8), ("C2", 1, 3), ("C2", 70000, 20)])
@pytest.mark.parametrize("packed", [True, False])
def test_steps_equal_oracle_batch_by_batch(hip, name, B, n, packed):
    prog, cfg = synth.config_program(name)
    nf = cfg["num_f"]
    hp = hip.HipProgram(prog)
    key = prng.key(77)
    # launch-plan feedback first (fused groups need "short hard-row lists" from earlier launches), on the same handle
    warm = [synth.synth_f(B, nf, cfg["p_bit"], seed=500 + i) for i in range(3)]
    _run_steps(hp, prog, warm, prng.key(1), nf, packed=packed)
    fs = [synth.synth_f(B, nf, cfg["p_bit"] * (1 + (i % 3)), seed=10 + i) for i in range(n)]
    outs, key_after = _run_steps(hp, prog, fs, key, nf, packed=packed)
    end_key, subs = _subkeys(key, n)
    assert key_after == (end_key[0] & 0xFFFFFFFF, end_key[1] & 0xFFFFFFFF)
    op = OC.OracleProgram(prog)
    check = range(n) if B <= 3000 else (0, n // 2, n - 1)
    for i in check:
        m = min(B, 4000)  # an oracle slice of the large batches; the rest is compared with the serial kernel path below
        want = op.sample_program(fs[i][:m], subs[i])
        np.testing.assert_array_equal(outs[i][:m], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i}")
    if B > 3000:
        for i in check:
            want = hp.sample_batch(fs[i], subs[i])[0]
            np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i} vs the serial API")
    hp.close()


def test_steps_use_the_fused_first_pass_and_split_calls_agree(hip):
    """The same 20 batches as one call, as calls of 7 + 1 + 12, and with the fused path switched off: identical rows;
    the profile counter shows that the fused kernel really ran."""
    import os

    prog, cfg = synth.config_program("C2")
    nf, B, n = cfg["num_f"], 20000, 20
    fs = [synth.synth_f(B, nf, 0.02 + 0.01 * (i % 4), seed=300 + i) for i in range(n)]
    key = prng.key(5)
    runs = []
    for calls, env in (([20], None), ([7, 1, 12], None), ([20], "0")):
        if env is not None:
            os.environ["TSIM_AMD_FUSED_STEPS"] = env
        # (shallow=0: the swap of the background-built default tables resets the launch-plan feedback, and the one or two
        # batches planned right behind it go batch by batch - same rows, but this test counts fused batches)
        os.environ["TSIM_AMD_TUNE"] = "shallow=0"
        try:
            prog2, _ = synth.config_program("C2")
            hp = hip.HipProgram(prog2)
        finally:
            os.environ.pop("TSIM_AMD_FUSED_STEPS", None)
            os.environ.pop("TSIM_AMD_TUNE", None)
        _run_steps(hp, prog2, fs[:3], prng.key(1), nf, packed=True)  # feedback
        hp.profile_set_sampling(1)
        hp.profile_enable(2)
        hp.profile_read(reset=True)
        hp.profile_read_steps()
        outs, _ = _run_steps(hp, prog2, fs, key, nf, packed=True, calls=calls)
        _, launches = hp.profile_read(reset=True)
        steps = hp.profile_read_steps()
        hp.profile_enable(False)
        runs.append(outs)
        if env is None:
            assert steps == n and launches < n, (steps, launches)  # several batches per first-pass launch
        else:
            assert steps == 0 and launches == n
        hp.close()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["C2", "C4"])
def test_hard_rows_on_the_batch_lane_and_on_the_group_lane_agree(hip, name):
    """Fused groups of at most hard_inline_rows (TSIM_AMD_TUNE) shots (default: all) run their hard-row batch on their own
    first-pass lane, larger ones on the batch lane; a handle that sees both, in turn, returns the rows of a handle
    that only uses the batch lane - slots are handed from one stream to the other in both directions."""
    import os

    prog, cfg = synth.config_program(name)
    nf = cfg["num_f"]
    sizes = [4000, 4000, 300000, 4000, 300000, 300000, 4000]  # x 8 batches: <= 2e6 inline, 2.4e6 on the batch lane
    key = prng.key(91)
    runs = []
    for env in ("2000000", "0"):
        os.environ["TSIM_AMD_TUNE"] = "hard_inline_rows=" + env
        try:
            prog2, _ = synth.config_program(name)
            hp = hip.HipProgram(prog2)
        finally:
            os.environ.pop("TSIM_AMD_TUNE", None)
        outs = []
        k = key
        for r, B in enumerate(sizes):
            fs = [synth.synth_f(B, nf, cfg["p_bit"], seed=900 + 31 * r + i) for i in range(8)]
            if r == 0:
                _run_steps(hp, prog2, fs[:3], prng.key(1), nf, packed=True)  # feedback
            o, k2 = _run_steps(hp, prog2, fs, k, nf, packed=True)
            k = np.array(k2, dtype=np.uint32)
            outs += o
        runs.append(outs)
        hp.close()
    assert len(runs[0]) == len(runs[1]) == 8 * len(sizes)
    for a, b in zip(*runs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("B", [5000, 4097, 3])
def test_component_parallel_hard_rows_agree_with_one_block_per_row(hip, packed, B):
    """Programs of 2-4 components: the specialised first pass stores the hard rows too (direct outputs and the
    tabulated components' bits) and k_sample_hw evaluates the components that are left, one block per (row,
    component), ORing their bits into the row (32-bit atomics; compact rows whose byte range is not a whole number of
    words - B = 4097, 3 with 3-byte rows - fall back).  Same rows as one block per row (TSIM_AMD_TUNE=hard_comp_par=0) and as
    the oracle, normalisation-check row included."""
    import os

    runs = []
    for env in ("1", "0"):
        os.environ["TSIM_AMD_TUNE"] = "hard_comp_par=" + env
        try:
            prog, cfg = synth.config_program("C4")
            hp = hip.HipProgram(prog)
        finally:
            os.environ.pop("TSIM_AMD_TUNE", None)
        nf = cfg["num_f"]
        fs = [synth.synth_f(B, nf, 0.04, seed=70 + i) for i in range(11)]
        _run_steps(hp, prog, fs[:3], prng.key(1), nf, packed=packed)  # feedback: few hard rows -> the block-per-row kernel
        key = prng.key(77)
        outs, _ = _run_steps(hp, prog, fs, key, nf, packed=packed)
        runs.append(outs)
        if env == "1":
            op = OC.OracleProgram(prog)
            _, subs = _subkeys(key, len(fs))
            for i in (0, 5, 10):
                want = op.sample_program(fs[i], subs[i])
                np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i}")
        hp.close()
    for a, b in zip(*runs):
        np.testing.assert_array_equal(a, b)


def test_tables_deepen_when_the_hard_rows_are_too_many_for_the_block_per_row_kernel(hip):
    """C3 at weight-5 tables leaves ~340 hard rows per 10^6 shots: too many for k_sample_hw (more than
    hard_wave_rows per eight launches), far below the 1 % of the dense rule.  A handle that has launched
    deep_after (TSIM_AMD_TUNE) rows in that state (here: 1) builds the weight-6 tables - in the background, while the
    weight-5 tables keep serving; TSIM_AMD_DEEP_TABLES=-1 never does.  Same rows from both, before, during and after the build."""
    import os

    B, n = 400_000, 8
    runs, depths = [], []
    for env in ({"TSIM_AMD_TUNE": "deep_after=1,shallow=0"}, {"TSIM_AMD_DEEP_TABLES": "-1", "TSIM_AMD_TUNE": "shallow=0"}):
        os.environ.update(env)
        try:
            prog, cfg = synth.config_program("C3")
            hp = hip.HipProgram(prog)
        finally:
            for k in env:
                os.environ.pop(k, None)
        nf = cfg["num_f"]
        assert hp.info()["pattern_max_weight"] == [5]
        fs = [synth.synth_f(B, nf, cfg["p_bit"], seed=300 + i) for i in range(n)]
        outs = []
        key = prng.key(5)
        for call in range(6):
            o, k2 = _run_steps(hp, prog, fs, key, nf, packed=True)
            key = np.array(k2, dtype=np.uint32)
            outs += [o[0], o[n - 1]]
        # the deeper tables are built in the background, one slice per launch plan (tsim_tables.hip): keep sampling
        extra = 0
        while "deep_after" in str(env) and hp.info()["pattern_max_weight"] == [5] and extra < 600:
            _run_steps(hp, prog, fs[:2], key, nf, packed=True)
            extra += 1
        o, k2 = _run_steps(hp, prog, fs, key, nf, packed=True)
        outs += [o[0], o[n - 1]]
        depths.append(hp.info()["pattern_max_weight"])
        runs.append(outs)
        hp.close()
    assert depths == [[6], [5]]
    for a, b in zip(*runs):
        np.testing.assert_array_equal(a, b)


def test_steps_with_shot_offset_and_dense_rows(hip):
    """A shard (shot_offset > 0: no normalisation-check row) and batches dense enough that the launch plan leaves the
    fused path (many hard rows): still the oracle's bits."""
    prog, cfg = synth.config_program("C2")
    nf, B = cfg["num_f"], 6000
    hp = hip.HipProgram(prog)
    op = OC.OracleProgram(prog)
    for p_bit, off in ((0.02, 123457), (0.25, 0), (0.02, 0)):
        fs = [synth.synth_f(B, nf, p_bit, seed=40 + i) for i in range(10)]
        key = prng.key(31)
        outs, _ = _run_steps(hp, prog, fs, key, nf, packed=True, shot_offset=off)
        _, subs = _subkeys(key, len(fs))
        for i in (0, 4, 9):
            want = op.sample_program(fs[i], subs[i], shot_offset=off)
            np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"p_bit {p_bit} batch {i}")
    hp.close()


@pytest.mark.parametrize("name,dense,extra", [("C2", 0.3, ""), ("C2", 0.08, ""), ("C4", 0.2, ""), ("C2", 0.3, ",hard_wave=0"), ("C3", 0.25, ",hard_wave=0")])
def test_a_jump_of_the_noise_level_goes_through_the_overflow_grid_and_back(hip, name, dense, extra):
    """The launch plan follows the hard-row counts of EARLIER launches.  The first group after a jump from sparse to dense
    batches therefore hands lists of thousands of rows to kernels sized for ten: they take the head of every list,
    k_sample4_over (tsim_kernel4.hip.h) the rest - including, when it lies there, the row of the normalisation check
    (sampler.py:66-72).  Afterwards the plan must find its way back to the fused first pass (the probe launch of a dense
    phase).  Every phase against the oracle; the same rows with the overflow grid switched off (whole lists on the latency
    kernels)."""
    import os

    prog, cfg = synth.config_program(name)
    nf, B, n = cfg["num_f"], 40_000, 8
    op = OC.OracleProgram(prog)
    phases = [cfg["p_bit"], cfg["p_bit"], dense, dense, cfg["p_bit"], cfg["p_bit"], cfg["p_bit"], cfg["p_bit"]]
    results = {}
    for over in (1, 0):
        # (hard_wave=0: the jump group's lists go to k_sample4h_multi and ITS workers; shallow=0: the plan sequence this test
        # follows starts from the default table depth - the shallow start has its own test, test_gpu_pattern_tables.py)
        os.environ["TSIM_AMD_TUNE"] = f"hard_overflow={over}{extra},shallow=0"
        try:
            hp = hip.HipProgram(prog)
        finally:
            os.environ.pop("TSIM_AMD_TUNE", None)
        hp.profile_set_sampling(1)
        hp.profile_enable(2)
        key = prng.key(77)
        outs, fused_steps = [], []
        for ph, p_bit in enumerate(phases):
            fs = [synth.synth_f(B, nf, p_bit, seed=1000 * ph + i) for i in range(n)]
            hp.profile_read_steps()
            dv = []
            o, k2 = _run_steps(hp, prog, fs, key, nf, packed=True, devs=dv)
            fused_steps.append(hp.profile_read_steps())
            if over == 1:
                _, subs = _subkeys(key, n)
                for i in (0, n - 1):
                    want, wdev = op.sample_program(fs[i], subs[i], return_devs=True)
                    np.testing.assert_array_equal(o[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"phase {ph} (p_bit {p_bit}) batch {i}")
                    # (the check row sits wherever the atomics put it in its list - in a dense batch mostly behind the latency kernel's share)
                    np.testing.assert_array_equal(dv[i], np.asarray(wdev, np.float32), err_msg=f"normalisation deviation, phase {ph} batch {i}")
            key = np.array(k2, dtype=np.uint32)
            outs.append(o)
        hp.profile_enable(False)
        hp.close()
        results[over] = outs
        assert fused_steps[1] == n and fused_steps[2] == n, f"the jump was not taken by a fused group: {fused_steps}"
        assert fused_steps[-1] == n, f"the plan did not return to the fused first pass: {fused_steps}"
    for a, b in zip(results[1], results[0]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("num_f,n_direct,B,n", [(24, 24, 5000, 11), (64, 40, 70001, 9), (128, 128, 3000, 3), (100, 1, 257, 17), (7, 7, 1, 2)])
@pytest.mark.parametrize("packed", [True, False])
def test_programs_without_components_stream_through_the_direct_kernel(hip, num_f, n_direct, B, n, packed):
    """Clifford-only programs (every output direct: out = f[idx] ^ flip, sampler.py:140-145) through the several-batches
    call: groups of up to eight batches as one streaming grid (tsim_direct.hip.h) - shuffled output order, flips, both
    output layouts, row counts that are no multiple of the block; the key state advances as for any other program."""
    if num_f == 24:
        prog, cfg = synth.config_program("C1")
    else:
        prog = synth.synth_program(num_f=num_f, n_direct=n_direct, components=[], seed=11 * num_f + n_direct, shuffle_outputs=True,
                                   direct_flip_fraction=0.4)
    hp = hip.HipProgram(prog)
    key = prng.key(3)
    fs = [synth.synth_f(B, num_f, 0.3, seed=70 + i) for i in range(n)]
    outs, key_after = _run_steps(hp, prog, fs, key, num_f, packed=packed)
    end_key, subs = _subkeys(key, n)
    assert key_after == (end_key[0] & 0xFFFFFFFF, end_key[1] & 0xFFFFFFFF)
    op = OC.OracleProgram(prog)
    for i in range(n):
        want = op.sample_program(fs[i], subs[i])
        np.testing.assert_array_equal(outs[i], np.packbits(want, axis=1, bitorder="little"), err_msg=f"batch {i}")
    assert hp.info()["n_components"] == 0
    hp.close()
