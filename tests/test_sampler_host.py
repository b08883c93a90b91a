"""Host orchestration of tsim_amd.sampler (CPU): the HIP call is replaced by the oracle.

Mirrors the technique of the reference's own unit tests, which spy on / replace
``tsim.sampler.sample_program`` (test/unit/test_postselection.py:186-274) and patch
``_estimate_batch_size`` / ``ChannelSampler.sample`` (test/unit/test_sampler.py:247-346).
"""

from unittest.mock import patch

import numpy as np
import pytest

import tsim_amd.sampler as sampler_module
from oracle import oracle_np as O
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.program import CompiledComponent, make_program, scalar_graphs_from_terms
from tsim_amd.sampler import CompiledDetectorSampler, CompiledMeasurementSampler


def oracle_sample_program(program, f_params, key):
    return O.sample_program(program, np.asarray(f_params), key)


@pytest.fixture(autouse=True)
def use_oracle_backend(monkeypatch):
    monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)


NO_NOISE = dict(channel_probs=[], error_transform=np.zeros((0, 0), np.uint8))


def random_bit_component(output_index, f_index=None):
    """One uniformly random output; optionally 'depends' on one f bit (F = 1)."""
    F = 0 if f_index is None else 1
    lv0 = scalar_graphs_from_terms(F, [dict()])
    lv1 = scalar_graphs_from_terms(F + 1, [dict(power2=-1)])
    fsel = np.zeros(0, np.int32) if f_index is None else np.asarray([f_index], np.int32)
    return CompiledComponent((output_index,), fsel, (lv0, lv1))


def det_obs_sampler(seed=None):
    """R 0 1 2; X 2; M 0 1 2; DETECTOR rec[-2]; DETECTOR rec[-3]; OBSERVABLE_INCLUDE(0) rec[-1].

    No noise: three deterministic outputs (0, 0, 1) as compiled single-output components.
    """
    comps = [synth.single_output_component(0, zero=True), synth.single_output_component(1, zero=True),
             synth.single_output_component(2, one=True)]
    prog = make_program(comps, [], 3, 2)
    return CompiledDetectorSampler(prog, seed=seed, **NO_NOISE)


def mixed_sampler(p=0.5, seed=0, flip=False, with_obs=False):
    """X_ERROR(p) 0; R 1; H 1; M 0 1; DETECTOR rec[-2]; DETECTOR rec[-1] rec[-2] [; OBS rec[-1]].

    det0 is direct (f0, optionally flipped by an X gate), det1 a compiled random bit.
    """
    comps = [random_bit_component(1, f_index=0)]
    n_out = 2
    if with_obs:
        comps.append(random_bit_component(2))
        n_out = 3
    prog = make_program(comps, [(0, 0, flip)], n_out, 2)
    return CompiledDetectorSampler(
        prog, channel_probs=[error_probs(p)], error_transform=np.array([[1]], np.uint8), seed=seed
    )


# ---- output arrangement (test/unit/test_sampler.py:9-67) -----------------------


def test_detector_sampler_args():
    s = det_obs_sampler()
    assert np.array_equal(s.sample(1), [[0, 0]])
    assert np.array_equal(s.sample(1, append_observables=True), [[0, 0, 1]])
    assert np.array_equal(s.sample(1, prepend_observables=True), [[1, 0, 0]])
    assert np.array_equal(s.sample(1, prepend_observables=True, append_observables=True), [[1, 0, 0, 1]])
    d, o = s.sample(1, separate_observables=True)
    assert np.array_equal(d, [[0, 0]]) and np.array_equal(o, [[1]])


@pytest.mark.parametrize(
    "kwargs",
    [
        {"separate_observables": True, "append_observables": True},
        {"separate_observables": True, "prepend_observables": True},
        {"separate_observables": True, "append_observables": True, "prepend_observables": True},
    ],
)
def test_invalid_flag_combos_raise(kwargs):
    with pytest.raises(ValueError, match="separate_observables"):
        det_obs_sampler().sample(1, **kwargs)


def test_zero_shots_and_dtypes():
    """test/unit/test_sampler.py:70-177 shapes and dtypes."""
    s = det_obs_sampler()
    assert s.sample(0).shape == (0, 2) and s.sample(0).dtype == np.bool_
    d, o = s.sample(0, separate_observables=True)
    assert d.shape == (0, 2) and o.shape == (0, 1)
    assert s.sample(0, bit_packed=True).shape == (0, 1)
    assert s.sample(5, bit_packed=True).dtype == np.uint8
    d, ref = s._sample_batches(0, compute_reference=True)
    assert d.shape == (0, 3) and ref.shape == (3,)
    empty = CompiledMeasurementSampler(make_program([], [], 0, 0), **NO_NOISE)
    assert empty.sample(7).shape == (7, 0)


def test_argument_validation():
    s = det_obs_sampler()
    with pytest.raises(ValueError, match="shots must be non-negative"):
        s.sample(-1)
    with pytest.raises(ValueError, match="batch_size must be at least 1"):
        s.sample(1, batch_size=0)


# ---- seeded KATs through the sampler classes -------------------------------------


def test_seed_kat_through_sampler_class():
    """test/unit/test_sampler.py:223-233 with the full host path (key per batch, auto batch)."""
    for _ in range(2):
        s = CompiledMeasurementSampler(synth.kat_h_m(), seed=0, **NO_NOISE)
        assert [int(np.count_nonzero(s.sample(100))) for _ in range(4)] == [48, 53, 52, 50]


def test_r_gate_kat_and_detector_variant():
    """test/integration/test_sampler_circuits.py:90-109."""
    m = CompiledMeasurementSampler(synth.kat_r_gate(), seed=0, **NO_NOISE).sample(10)
    assert [int(np.count_nonzero(m[:, i])) for i in range(3)] == [7, 4, 0]
    det = make_program([random_bit_component(0)], [], 1, 1)
    d = CompiledDetectorSampler(det, seed=0, **NO_NOISE).sample(10)
    assert np.count_nonzero(d) == 7


def test_x_error_detector_kat_direct_path():
    """test/integration/test_sampler_circuits.py:25-37: fully direct, seed=1 -> 4 of 10."""
    s = CompiledDetectorSampler(
        synth.kat_x_error_detector(), channel_probs=[error_probs(0.3)],
        error_transform=np.array([[1]], np.uint8), seed=1,
    )
    with patch.object(sampler_module, "sample_program", side_effect=AssertionError("device used")):
        d = s.sample(10)
    assert np.count_nonzero(d) == 4


# ---- batching (test/unit/test_sampler.py:247-346) ---------------------------------


@pytest.mark.parametrize(("shots", "expected_bs"), [(100, 25), (101, 26)])
def test_auto_batch(shots, expected_bs):
    s = CompiledMeasurementSampler(synth.kat_h_m(), seed=42, **NO_NOISE)
    with (
        patch.object(type(s), "_estimate_batch_size", return_value=30),
        patch.object(s._channel_sampler, "sample", wraps=s._channel_sampler.sample) as spy,
    ):
        out = s.sample(shots)
    assert out.shape == (shots, 1)
    assert [c.args[0] for c in spy.call_args_list] == [expected_bs] * 4


@pytest.mark.parametrize(
    ("shots", "max_batch", "batch_size", "compute_ref", "expected_bs", "expected_n"),
    [
        (99, 30, None, True, 25, 4),
        (100, 30, None, True, 26, 4),
        (200, 30, None, True, 29, 7),
        (100, 30, None, False, 25, 4),
        (100, None, 50, True, 51, 2),
        (100, None, 51, True, 51, 2),
        (2, None, 1, True, 2, 2),
        (12, None, 3, True, 4, 4),
    ],
)
def test_batch_size_with_reference(shots, max_batch, batch_size, compute_ref, expected_bs, expected_n):
    s = CompiledMeasurementSampler(synth.kat_h_m(), seed=0, **NO_NOISE)
    with (
        patch.object(type(s), "_estimate_batch_size", return_value=max_batch or 9999),
        patch.object(s._channel_sampler, "sample", wraps=s._channel_sampler.sample) as spy,
    ):
        res = s._sample_batches(shots, batch_size=batch_size, compute_reference=compute_ref)
    if compute_ref:
        samples, ref = res
        assert samples.shape == (shots, 1) and ref.shape == (1,)
    else:
        assert res.shape == (shots, 1)
    sizes = [c.args[0] for c in spy.call_args_list]
    assert all(b == expected_bs for b in sizes) and len(sizes) == expected_n


# ---- reference sample (test/unit/test_sampler.py:349-451) --------------------------


def test_reference_sample_xor_variants():
    s = det_obs_sampler()
    d = s.sample(1, append_observables=True, use_detector_reference_sample=True, use_observable_reference_sample=True)
    assert not d.any()
    d, o = det_obs_sampler().sample(1, separate_observables=True, use_detector_reference_sample=True)
    assert not d.any() and np.array_equal(o, [[1]])
    d, o = det_obs_sampler().sample(1, separate_observables=True, use_observable_reference_sample=True)
    assert np.array_equal(d, [[0, 0]]) and not o.any()
    packed = det_obs_sampler().sample(
        1, append_observables=True, bit_packed=True,
        use_detector_reference_sample=True, use_observable_reference_sample=True,
    )
    assert np.array_equal(packed, np.packbits(np.zeros(3, bool), bitorder="little").reshape(1, -1))
    a = det_obs_sampler(seed=3).sample(1, append_observables=True)
    b = det_obs_sampler(seed=3).sample(
        1, append_observables=True, use_detector_reference_sample=False, use_observable_reference_sample=False
    )
    assert np.array_equal(a, b)


def test_reference_row_is_noiseless_and_stripped():
    """The first batch's row 0 is evaluated with f = 0 and removed (sampler.py:395-404)."""
    s = mixed_sampler(p=1.0, seed=0, flip=True)  # det0 = f0 ^ 1: fires only for the noiseless reference
    out, ref = s._sample_batches(8, batch_size=4, compute_reference=True)
    assert out.shape == (8, 2) and ref[0] and not out[:, 0].any()


# ---- post-selection (test/unit/test_postselection.py) ------------------------------


def spy_rows(monkeypatch):
    rows = []

    def spy(program, f_params, key):
        rows.append(f_params.shape[0])
        return oracle_sample_program(program, f_params, key)

    monkeypatch.setattr(sampler_module, "sample_program", spy)
    return rows


def test_postselection_validation():
    s = mixed_sampler()
    with pytest.raises(ValueError, match="postselection_mask must have shape"):
        s.sample(1, postselection_mask=np.array([True, False, False]))
    with pytest.raises(ValueError, match="postselection_mask must have shape"):
        s.sample(1, postselection_mask=np.zeros((2, 1), bool))
    with pytest.raises(ValueError, match="shots must be non-negative"):
        s.sample(-1, postselection_mask=np.array([True, False]))
    with pytest.raises(ValueError, match="batch_size must be at least 1"):
        s.sample(1, batch_size=0, postselection_mask=np.array([True, False]))


def test_postselection_none_and_all_false_match_default():
    a = mixed_sampler(seed=5).sample(16, batch_size=4)
    b = mixed_sampler(seed=5).sample(16, batch_size=4, postselection_mask=None)
    c = mixed_sampler(seed=5).sample(16, batch_size=4, postselection_mask=np.zeros(2, bool))
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_postselection_shapes():
    s = mixed_sampler()
    mask = np.array([True, False])
    assert s.sample(0, postselection_mask=mask).shape == (0, 2)
    assert s.sample(1, postselection_mask=mask).shape == (1, 2)
    assert s.sample(17, batch_size=4, postselection_mask=mask).shape == (17, 2)


def test_postselection_all_discarded_never_calls_device(monkeypatch):
    rows = spy_rows(monkeypatch)
    out = mixed_sampler(p=1.0).sample(20, batch_size=4, postselection_mask=np.array([True, False]))
    assert rows == [] and out[:, 0].all() and not out[:, 1].any()


def test_postselection_partial_discards(monkeypatch):
    rows = spy_rows(monkeypatch)
    s = mixed_sampler(seed=2)
    mask = np.array([True, False])
    out = s.sample(64, batch_size=8, postselection_mask=mask)
    discarded = out[:, 0] & mask[0]
    assert discarded.any() and (~discarded).any()
    assert not out[discarded, 1].any()
    assert sum(rows) < 64 + 8 and sum(rows) >= int((~discarded).sum())
    assert all(r == 8 for r in rows), rows  # fixed batch shape, final batch padded


def test_postselection_non_direct_mask_runs_everything(monkeypatch):
    rows = spy_rows(monkeypatch)
    mixed_sampler(seed=9).sample(16, batch_size=8, postselection_mask=np.array([False, True]))
    assert sum(rows) == 16


def test_postselection_direct_cols_equal_numpy():
    s = mixed_sampler(seed=3)
    drawn = []
    original = s._channel_sampler.sample

    def capture(n):
        b = original(n)
        drawn.append(b.copy())
        return b

    with patch.object(s._channel_sampler, "sample", side_effect=capture):
        out = s.sample(8, batch_size=4, postselection_mask=np.array([True, False]))
    expect = s._compute_direct_outputs(np.concatenate(drawn))
    dd = s._direct_detector_mask
    assert np.array_equal(out[:, :2] & dd, expect[:, :2] & dd)


def test_postselection_detector_reference_cancels_discard(monkeypatch):
    """With X before X_ERROR the raw det0 fires when f0 = 0; XOR with the reference flips that."""
    mask = np.array([True, False])
    without = spy_rows(monkeypatch)
    mixed_sampler(seed=0, flip=True).sample(32, batch_size=8, postselection_mask=mask)
    n_without = sum(without)
    with_ref = spy_rows(monkeypatch)
    out = mixed_sampler(seed=0, flip=True).sample(
        32, batch_size=8, postselection_mask=mask, use_detector_reference_sample=True
    )
    # reference det0 = 1, so survivors are the rows where f0 = 0 ... both halves are ~50%;
    # the reference computation itself adds one extra 1-row device call
    assert with_ref[0] == 1 and out.shape == (32, 2)  # the reference sample is its own 1-row call
    assert n_without > 0 and sum(with_ref) > 1


def test_postselection_observable_reference_only_on_device_rows():
    s = mixed_sampler(p=0.5, seed=4, with_obs=True)
    mask = np.array([True, False])
    d, o = s.sample(40, batch_size=8, postselection_mask=mask, separate_observables=True,
                    use_observable_reference_sample=True)
    assert d.shape == (40, 2) and o.shape == (40, 1)
    discarded = d[:, 0]
    assert not o[discarded].any()  # rows that never ran keep False observables


def test_repr_has_statistics():
    r = repr(mixed_sampler())
    assert r.startswith("CompiledDetectorSampler(1 direct, 2 graphs, 1 error channel bits")
