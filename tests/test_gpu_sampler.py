"""End-to-end sampler classes on the MI355X (real HIP backend), vs oracle-driven twins."""

import numpy as np
import pytest

import tsim_amd.sampler as sampler_module
from oracle import oracle_np as O
from tsim_amd import synth
from tsim_amd.channels import error_probs, pauli_channel_1_probs
from tsim_amd.program import CompiledComponent, make_program, scalar_graphs_from_terms
from tsim_amd.sampler import CompiledDetectorSampler, CompiledMeasurementSampler, CompiledStateProbs

pytestmark = pytest.mark.gpu

NO_NOISE = dict(channel_probs=[], error_transform=np.zeros((0, 0), np.uint8))


def oracle_sample_program(program, f_params, key):
    return O.sample_program(program, np.asarray(f_params), key)


def test_seed_kat_through_hip_sampler(hip):
    s = CompiledMeasurementSampler(synth.kat_h_m(), seed=0, **NO_NOISE)
    assert [int(np.count_nonzero(s.sample(100))) for _ in range(4)] == [48, 53, 52, 50]
    s = CompiledMeasurementSampler(synth.kat_t_gate(), seed=0, **NO_NOISE)
    assert int(np.count_nonzero(s.sample(100))) == 9


def noisy_program():
    """Two noisy-T components + two direct detectors driven by a small Pauli noise model."""
    comps = [synth.noisy_t_component(2, 1), synth.noisy_t_component(3, 3)]
    prog = make_program(comps, [(0, 0, False), (1, 2, True)], 4, 2)
    probs = [error_probs(0.2), pauli_channel_1_probs(0.05, 0.1, 0.15), error_probs(0.3)]
    T = np.array([[1, 0, 0, 0], [0, 1, 0, 1], [0, 0, 1, 0], [1, 0, 1, 1]], np.uint8)
    return prog, dict(channel_probs=probs, error_transform=T)


@pytest.mark.parametrize("kwargs", [
    dict(),
    dict(append_observables=True, bit_packed=True),
    dict(separate_observables=True, use_detector_reference_sample=True, use_observable_reference_sample=True),
    dict(append_observables=True, postselection_mask=np.array([True, False])),
    dict(prepend_observables=True, postselection_mask=np.array([True, True]), use_detector_reference_sample=True),
])
@pytest.mark.parametrize("mode", ["auto", "rows", "faithful"])
def test_detector_sampler_matches_oracle_twin(hip, monkeypatch, kwargs, mode):
    """Same seed, same batch size: the HIP-backed sampler equals the oracle-backed one bit for bit."""
    prog, noise = noisy_program()
    a = CompiledDetectorSampler(prog, seed=11, mode=mode, **noise).sample(1000, batch_size=300, **kwargs)
    monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)
    b = CompiledDetectorSampler(prog, seed=11, **noise).sample(1000, batch_size=300, **kwargs)
    if isinstance(a, tuple):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    else:
        assert np.array_equal(a, b)


def test_noisy_t_statistics(hip):
    """Physically consistent program: P(m=1 | f) = sin^2(pi/8) or cos^2(pi/8); 4 sigma check."""
    prog = make_program([synth.noisy_t_component(0, 0)], [], 1, 0)
    s = CompiledMeasurementSampler(prog, channel_probs=[error_probs(0.25)], error_transform=np.array([[1]], np.uint8), seed=5)
    n = 200_000
    m = s.sample(n, batch_size=50_000)[:, 0]
    p = 0.75 * np.sin(np.pi / 8) ** 2 + 0.25 * np.cos(np.pi / 8) ** 2
    assert abs(m.mean() - p) < 4 * np.sqrt(p * (1 - p) / n)


def test_state_probs_matches_oracle(hip):
    """CompiledStateProbs.probability_of (sampler.py:906-953) on a joint-mode program."""
    rng = np.random.default_rng(3)
    F, n = 6, 3
    lv0 = synth.synth_level(rng, F, 4, ta=(0, 4), tb=(0, 4), tc=(0, 4), td=(0, 2))
    lv1 = synth.synth_level(rng, F + n, 6, ta=(0, 4), tb=(0, 4), tc=(0, 4), td=(0, 2))
    comp = CompiledComponent((1, 2, 3), np.array([0, 2, 3, 4, 6, 7], np.int32), (lv0, lv1))
    prog = make_program([comp], [(0, 1, True)], 4, 0)
    probs = [error_probs(0.3)] * 8
    T = np.eye(8, dtype=np.uint8)
    state = np.array([1, 0, 1, 1], np.uint8)
    sp = CompiledStateProbs(prog, channel_probs=probs, error_transform=T, seed=2)
    got = sp.probability_of(state, batch_size=64)
    # oracle twin: same channel stream
    from tsim_amd.channels import ChannelSampler

    cs = ChannelSampler(probs, T, seed=int(np.random.default_rng(2).integers(0, 2**30)))
    f = cs.sample(64)
    fsel = f[:, comp.f_selection]
    pn = O.complex_abs(O.evaluate(lv0, fsel))
    pj = O.complex_abs(O.evaluate(lv1, np.hstack([fsel, np.tile(state[[1, 2, 3]], (64, 1))])))
    pj = pj * ((f[:, 1].astype(bool) ^ True) == bool(state[0]))
    with np.errstate(divide="ignore", invalid="ignore"):
        want = (pj / pn).astype(np.float32)
    np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError, match="state must have shape"):
        sp.probability_of(np.zeros(3), batch_size=4)
    with pytest.raises(ValueError, match="batch_size must be at least 1"):
        sp.probability_of(state, batch_size=0)


def test_evaluate_seam_function(hip):
    """tsim_amd.backend.evaluate(circuit, param_vals) mirrors compile/evaluate.py:16."""
    rng = np.random.default_rng(8)
    lv = synth.synth_level(rng, 20, 5)
    pv = (rng.random((100, 20)) < 0.5).astype(np.uint8)
    z = hip.evaluate(lv, pv)
    assert z.dtype == np.complex64
    np.testing.assert_array_equal(z.view(np.float32), O.evaluate(lv, pv).view(np.float32))
    from tsim_amd.program import empty_scalar_graphs

    assert np.array_equal(hip.evaluate(empty_scalar_graphs(2), np.ones((3, 2), np.uint8)), np.zeros(3, complex))


def test_vanishing_marginal_raises_and_warns(hip):
    """sampler.py:149-161: deviation ~1 -> ValueError; > 1e-5 -> warning."""
    bad = CompiledComponent(
        (0,), np.zeros(0, np.int32),
        (scalar_graphs_from_terms(0, [dict()]), scalar_graphs_from_terms(1, [dict(power2=-40)])),
    )
    # p0 + p1 = 2^-40 + 2^-40 -> norm ~ 0 -> |norm - 1| ~ 1
    prog = make_program([bad], [], 1, 0)
    with pytest.raises(ValueError, match="vanishing marginal"):
        hip.sample_program(prog, np.zeros((4, 0), np.uint8), (1, 2))
    off = CompiledComponent(
        (0,), np.zeros(0, np.int32),
        (scalar_graphs_from_terms(0, [dict()]), scalar_graphs_from_terms(1, [dict(floatfactor=(3, 0, 0, 0), power2=-3)])),
    )
    with pytest.warns(UserWarning, match="not normalized"):
        hip.sample_program(make_program([off], [], 1, 0), np.zeros((4, 0), np.uint8), (1, 2))


def test_device_noise_pipeline_statistics(hip):
    """noise="device": the whole pipeline stays on the GPU; statistics equal the host-noise sampler's."""
    prog, noise = noisy_program()
    n = 400_000
    a = CompiledDetectorSampler(prog, seed=3, noise="device", **noise).sample(n, batch_size=100_000, append_observables=True)
    b = CompiledDetectorSampler(prog, seed=3, noise="host", **noise).sample(n, batch_size=100_000, append_observables=True)
    assert a.shape == b.shape == (n, 4) and a.dtype == np.bool_
    pa, pb = a.mean(axis=0), b.mean(axis=0)
    sigma = np.sqrt(np.maximum(pb * (1 - pb), 1e-9) * 2 / n)
    assert np.all(np.abs(pa - pb) < 6 * sigma + 1e-6), (pa, pb)
    # correlations between a direct detector and an observable it shares an error with
    ca, cb = (a[:, 1] & a[:, 3]).mean(), (b[:, 1] & b[:, 3]).mean()
    assert abs(ca - cb) < 6 * np.sqrt(max(cb * (1 - cb), 1e-9) * 2 / n) + 1e-6
    # reproducible for a fixed seed, reference-sample flags work
    a2 = CompiledDetectorSampler(prog, seed=3, noise="device", **noise).sample(n, batch_size=100_000, append_observables=True)
    assert np.array_equal(a, a2)
    d, o = CompiledDetectorSampler(prog, seed=3, noise="device", **noise).sample(
        1000, separate_observables=True, use_detector_reference_sample=True)
    assert d.shape == (1000, 2) and o.shape == (1000, 2)
    with pytest.raises(ValueError):
        CompiledDetectorSampler(prog, seed=3, noise="gpu", **noise)


@pytest.mark.parametrize("mode", ["auto", "rows", "faithful"])
def test_device_postselection(hip, mode):
    """noise="device" + postselection_mask: discarded rows never reach the sampling kernel.

    Checks the reference's contract (sampler.py:776-781): all rows returned; a discarded row keeps
    its direct detector columns and has False in every compiled column; survivors are complete; the
    statistics of the survivors equal those of the host path.
    """
    prog, noise = noisy_program()  # det0 = f0 (direct), det1 = f2 ^ 1 (direct), obs 2, 3 compiled
    n = 200_000
    mask = np.array([True, False])
    dev = CompiledDetectorSampler(prog, seed=8, noise="device", mode=mode, **noise)
    a = dev.sample(n, batch_size=64_000, append_observables=True, postselection_mask=mask)
    assert a.shape == (n, 4)
    disc = a[:, 0] & mask[0]
    assert disc.any() and (~disc).any()
    assert not a[disc, 2:].any()  # compiled columns of discarded rows are False
    host = CompiledDetectorSampler(prog, seed=8, noise="host", **noise)
    b = host.sample(n, batch_size=64_000, append_observables=True, postselection_mask=mask)
    sa, sb = a[~disc], b[~(b[:, 0] & mask[0])]
    pa, pb = sa.mean(axis=0), sb.mean(axis=0)
    sig = np.sqrt(np.maximum(pb * (1 - pb), 1e-9) * (1 / len(sa) + 1 / len(sb)))
    assert np.all(np.abs(pa - pb) < 6 * sig + 1e-6), (pa, pb)
    assert abs(disc.mean() - 0.2) < 0.01  # P(f0 = 1) = 0.2 in the noise model
    # with the detector reference XOR: det1's reference value is 1, so it reads f2 afterwards
    c = CompiledDetectorSampler(prog, seed=8, noise="device", **noise).sample(
        50_000, batch_size=20_000, postselection_mask=np.array([False, True]), use_detector_reference_sample=True)
    assert c.shape == (50_000, 2) and 0.05 < c[:, 1].mean() < 0.6
    # a mask that discards everything: nothing is sampled, compiled columns all False
    alld = CompiledDetectorSampler(prog, seed=8, noise="device", channel_probs=[error_probs(1.0)] + noise["channel_probs"][1:],
                                   error_transform=noise["error_transform"])
    z = alld.sample(5000, batch_size=2000, append_observables=True, postselection_mask=mask)
    assert z[:, 0].all() and not z[:, 2:].any()


@pytest.mark.parametrize("ref_flags", [{}, {"use_detector_reference_sample": True}, {"use_detector_reference_sample": True, "use_observable_reference_sample": True}])
def test_device_postselection_is_the_plain_result_with_discarded_rows_blanked(hip, ref_flags):
    """noise="device": every row goes through the fast sampling path and a device kernel blanks the rows in which a masked
    direct detector fires (tsim_postselect_rows_device).  A sampler with the same seed and no mask draws the same noise
    and the same bits, so the post-selected result must be exactly that result with the reference's rules applied
    (sampler.py:532-540: a discarded row keeps its direct detector columns, False elsewhere) - as bools, as bit_packed
    rows, with the reference-sample flags, over several batches and a ragged last one."""
    from tsim_amd import synth
    from tsim_amd.channels import error_probs

    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    kw = dict(channel_probs=[error_probs(0.03)] * nf, error_transform=np.eye(nf, dtype=np.uint8), noise="device")
    n, bs = 250_001, 100_000
    mk = lambda: CompiledDetectorSampler(prog, seed=21, **kw)  # noqa: E731
    s0 = mk()
    nd = s0._num_detectors
    direct = s0._direct_detector_mask
    mask = np.zeros(nd, dtype=bool)
    mask[np.flatnonzero(direct)[::2]] = True  # every other directly readable detector
    sp = mk()  # (the reference sample is computed first and takes a key: the twin does the same before it samples)
    ref = sp._compute_reference_sample() if ref_flags else None
    plain = sp.sample(n, batch_size=bs, append_observables=True)
    if ref is None:
        ref = np.zeros(plain.shape[1], dtype=bool)
    got = mk().sample(n, batch_size=bs, append_observables=True, postselection_mask=mask, **ref_flags)
    det_ref = ref[:nd] if ref_flags.get("use_detector_reference_sample") else np.zeros(nd, dtype=bool)
    obs_ref = ref[nd:] if ref_flags.get("use_observable_reference_sample") else np.zeros(plain.shape[1] - nd, dtype=bool)
    gone = ((plain[:, :nd] ^ det_ref) & mask).any(axis=1)
    assert 0.05 < gone.mean() < 0.95
    want = plain.copy()
    want[gone, nd:] = False
    want[np.ix_(gone, np.flatnonzero(~direct))] = False
    want[~gone, :nd] ^= det_ref
    want[gone, :nd] ^= det_ref & direct
    want[~gone, nd:] ^= obs_ref
    np.testing.assert_array_equal(got, want)
    packed = mk().sample(n, batch_size=bs, append_observables=True, postselection_mask=mask, bit_packed=True, **ref_flags)
    np.testing.assert_array_equal(packed, np.packbits(want, axis=1, bitorder="little"))
    det_only = mk().sample(n, batch_size=bs, postselection_mask=mask, bit_packed=True, **ref_flags)
    np.testing.assert_array_equal(det_only, np.packbits(want[:, :nd], axis=1, bitorder="little"))


@pytest.mark.parametrize("ref_flags", [{}, {"use_detector_reference_sample": True, "use_observable_reference_sample": True}])
def test_host_noise_postselection_packed_rows_equal_the_bool_rows(hip, ref_flags):
    """noise="host" (the reference's stream and its compacted survivor batches, sampler.py:422-545): with bit_packed=True
    the blanking, the reference bits and the packing happen on the device (tsim_postselect_rows_device +
    tsim_compact_rows_device); the bytes must be np.packbits of what the bool path returns for the same seed."""
    from tsim_amd import synth
    from tsim_amd.channels import error_probs

    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    kw = dict(channel_probs=[error_probs(0.03)] * nf, error_transform=np.eye(nf, dtype=np.uint8), noise="host")
    mk = lambda: CompiledDetectorSampler(prog, seed=5, **kw)  # noqa: E731
    s0 = mk()
    nd = s0._num_detectors
    mask = np.zeros(nd, dtype=bool)
    mask[np.flatnonzero(s0._direct_detector_mask)[::3]] = True
    n, bs = 70_001, 30_000
    bools = mk().sample(n, batch_size=bs, append_observables=True, postselection_mask=mask, **ref_flags)
    assert 0.02 < (~bools[:, nd:].any(axis=1)).mean() < 0.98
    packed = mk().sample(n, batch_size=bs, append_observables=True, postselection_mask=mask, bit_packed=True, **ref_flags)
    np.testing.assert_array_equal(packed, np.packbits(bools, axis=1, bitorder="little"))
    det_only = mk().sample(n, batch_size=bs, postselection_mask=mask, bit_packed=True, **ref_flags)
    np.testing.assert_array_equal(det_only, np.packbits(bools[:, :nd], axis=1, bitorder="little"))


@pytest.mark.parametrize("noise", ["host", "device"])
def test_output_arrangement_on_the_device_equals_the_numpy_epilogue(hip, noise):
    """prepend / append / separate observables, the reference-sample flips and bit_packed in every combination: the
    arrays sample() hands out are arranged on the device (tsim_arrange_rows_device); they must equal what the reference's
    numpy epilogue (sampler.py:850-868, 665-669) makes of the plain rows and the reference sample of an identically
    seeded twin - including the host-noise case, where the reference row rides as row 0 of the first batch."""
    from tsim_amd import synth
    from tsim_amd.channels import error_probs

    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    kw = dict(channel_probs=[error_probs(0.02)] * nf, error_transform=np.eye(nf, dtype=np.uint8), noise=noise)
    mk = lambda: CompiledDetectorSampler(prog, seed=33, **kw)  # noqa: E731
    n, bs = 130_001, 50_000
    nd = mk()._num_detectors
    plain_noref = mk()._sample_batches(n, bs)
    plain_ref, ref = mk()._sample_batches(n, bs, compute_reference=True)
    assert ref.any()
    for flags in ({"prepend_observables": True}, {"prepend_observables": True, "append_observables": True},
                  {"separate_observables": True}, {"use_detector_reference_sample": True},
                  {"use_observable_reference_sample": True, "append_observables": True},
                  {"use_detector_reference_sample": True, "use_observable_reference_sample": True, "separate_observables": True},
                  {"use_detector_reference_sample": True, "prepend_observables": True}):
        wants_ref = flags.get("use_detector_reference_sample") or flags.get("use_observable_reference_sample")
        rows = (plain_ref if wants_ref else plain_noref).copy()
        if flags.get("use_detector_reference_sample"):
            rows[:, :nd] ^= ref[:nd]
        if flags.get("use_observable_reference_sample"):
            rows[:, nd:] ^= ref[nd:]
        det, obs = rows[:, :nd], rows[:, nd:]
        if flags.get("separate_observables"):
            want = (det, obs)
        elif flags.get("prepend_observables"):
            want = (np.concatenate([obs, det] + ([obs] if flags.get("append_observables") else []), axis=1),)
        else:
            want = (rows if flags.get("append_observables") else det,)
        for packed in (False, True):
            got = mk().sample(n, batch_size=bs, bit_packed=packed, **flags)
            got = got if isinstance(got, tuple) else (got,)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g, np.packbits(w, axis=1, bitorder="little") if packed else w, err_msg=f"{flags} packed={packed}")


@pytest.mark.parametrize("nbits,B", [(20, 1001), (5, 7), (64, 300), (121, 4096), (8, 3)])
def test_compact_rows_equals_numpy_packbits(hip, nbits, B):
    """tsim_compact_rows_device == np.packbits(bits, axis=1, bitorder="little") (sampler.py:665-669)."""
    from tsim_amd import synth

    prog, _ = synth.config_program("C1")
    hp = hip.HipProgram(prog)
    rng = np.random.default_rng(nbits * 1000 + B)
    bits = rng.integers(0, 2, size=(B, nbits), dtype=np.uint8)
    wo = (nbits + 63) // 64
    padded = np.zeros((B, wo * 8), np.uint8)
    pk = np.packbits(bits, axis=1, bitorder="little")
    padded[:, : pk.shape[1]] = pk
    d_in, d_out = hp.malloc(B * wo * 8), hp.malloc(B * pk.shape[1] + 16)
    hp.h2d(d_in, padded)
    hp.compact_rows_device(d_in.ptr, B, nbits, d_out.ptr)  # in_words defaults to ceil(nbits/64)
    hp.synchronize()
    got = np.zeros((B, pk.shape[1]), np.uint8)
    hp.d2h(got, d_out)
    np.testing.assert_array_equal(got, pk)


@pytest.mark.parametrize("noise", ["device", "host"])
@pytest.mark.parametrize("append", [False, True])
def test_bit_packed_device_fast_path_equals_packbits(hip, append, noise):
    """bit_packed=True with device noise takes the compact-rows path: same bytes as packing the
    unpacked result on the host (sampler.py:665-669)."""
    from tsim_amd import synth

    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    probs = [np.array([0.98, 0.02]) for _ in range(nf)]
    et = np.eye(nf, dtype=np.uint8)
    a = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=5, noise=noise)
    b = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=5, noise=noise)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        packed = a.sample(5003, batch_size=2048, bit_packed=True, append_observables=append)
        plain = b.sample(5003, batch_size=2048, append_observables=append)
    np.testing.assert_array_equal(packed, np.packbits(plain, axis=1, bitorder="little"))


def test_host_noise_device_route_equals_oracle_twin_at_scale(hip, monkeypatch):
    """The product route (native PCG64 channel stream -> packed rows -> pipelined launches -> one download) against
    the seam route driven by the oracle with numpy's generator: same seed, same batch size -> same bits, with and
    without the reference row and under post-selection (gather / dense sampling / scatter on the device)."""
    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    probs = [error_probs(0.03)] * (nf - 2) + [pauli_channel_1_probs(0.01, 0.02, 0.005)]
    et = np.eye(nf, dtype=np.uint8)
    mask = np.zeros(15, bool)
    mask[[0, 3]] = True
    cases = [dict(append_observables=True),
             dict(separate_observables=True, use_detector_reference_sample=True, use_observable_reference_sample=True),
             dict(append_observables=True, postselection_mask=mask),
             dict(append_observables=True, postselection_mask=mask, use_detector_reference_sample=True)]
    got = [CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=21).sample(7001, batch_size=2000, **kw)
           for kw in cases]
    monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)
    for kw, g in zip(cases, got):
        s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=21)
        s._channel_sampler._native = None  # numpy's own generator on the twin
        w = s.sample(7001, batch_size=2000, **kw)
        if isinstance(g, tuple):
            assert all(np.array_equal(x, y) for x, y in zip(g, w))
        else:
            assert np.array_equal(g, w)


def test_wide_program_through_the_sampler_equals_oracle_twin(hip, monkeypatch):
    """The same twin check on C5 (200-parameter component: pattern tables -> sparse-column kernel -> row kernel behind
    the sampler's pipelined launches), with post-selection on direct detectors and the reference rows."""
    prog, cfg = synth.config_program("C5")
    nf = cfg["num_f"]
    probs = [error_probs(0.02)] * (nf - 2) + [pauli_channel_1_probs(0.01, 0.02, 0.005)]
    et = np.eye(nf, dtype=np.uint8)
    mask = np.zeros(prog.num_detectors, bool)
    mask[[1, 7]] = True
    cases = [dict(append_observables=True),
             dict(separate_observables=True, use_detector_reference_sample=True, use_observable_reference_sample=True),
             dict(append_observables=True, postselection_mask=mask)]
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = [CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=33).sample(6001, batch_size=2500, **kw)
               for kw in cases]
        monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)
        for kw, g in zip(cases, got):
            s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=33)
            s._channel_sampler._native = None  # numpy's own generator on the twin
            w = s.sample(6001, batch_size=2500, **kw)
            if isinstance(g, tuple):
                assert all(np.array_equal(x, y) for x, y in zip(g, w))
            else:
                assert np.array_equal(g, w)


def test_more_batches_than_lanes_keep_their_f_rows(hip, monkeypatch):
    """Host noise, many more batches than f buffers (one per pipeline lane), kernels much slower than uploads: the
    cultivation shape on the full kernel (pattern tables off), 40 batches of 500 rows.  The upload of batch b + 8 reuses
    the buffer batch b's kernels read; the guard is `tsim_pipeline_wait_slot` - `sample_batch_device_end` no longer waits
    once the copy stream has joined the slot (torn rows).  Against the oracle-driven twin, bit for bit."""
    prog, cfg = synth.config_program("C4")
    nf = cfg["num_f"]
    probs = [error_probs(0.2)] * nf  # dense rows: nothing is tabulated, every row walks 1024 graphs
    et = np.eye(nf, dtype=np.uint8)
    monkeypatch.setenv("TSIM_AMD_PATTERN_TABLES", "0")
    got = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=5).sample(20000, batch_size=500, append_observables=True)
    monkeypatch.delenv("TSIM_AMD_PATTERN_TABLES")
    again = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=5).sample(20000, batch_size=500, append_observables=True)
    assert np.array_equal(got, again)
    monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)
    s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=5)
    s._channel_sampler._native = None
    want = s.sample(1500, batch_size=500, append_observables=True)  # the oracle is slow on this shape: the first three batches
    assert np.array_equal(got[:1500], want)


def test_device_noise_ring_is_not_overwritten_before_it_is_sampled(hip):
    """noise="device", more batches than the 32 f buffers of the ring: the noise kernel of batch k + 32 must wait for
    batch k's sampling kernels (a non-consuming wait on the slot's completion event).  A torn ring shows as results that
    differ between identically seeded runs or from the run that never wraps the ring (its first batches)."""
    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    kw = dict(channel_probs=[error_probs(0.02)] * nf, error_transform=np.eye(nf, dtype=np.uint8), seed=9, noise="device")
    shots = 40 * (1 << 20)
    a = CompiledDetectorSampler(prog, **kw).sample(shots, batch_size=1 << 20, append_observables=True, bit_packed=True)
    b = CompiledDetectorSampler(prog, **kw).sample(shots, batch_size=1 << 20, append_observables=True, bit_packed=True)
    assert np.array_equal(a, b)
    c = CompiledDetectorSampler(prog, **kw).sample(8 << 20, batch_size=1 << 20, append_observables=True, bit_packed=True)
    assert np.array_equal(a[: 8 << 20], c)
