"""tsim_amd.channels against golden vectors produced by the reference's own ChannelSampler."""

import os

import numpy as np
import pytest

from tsim_amd import channels as ch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "channels_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_golden_cases_bit_exact(gold):
    for ci in range(int(gold["n_cases"])):
        probs = [gold[f"c{ci}_probs{i}"] for i in range(int(gold[f"c{ci}_n_channels"]))]
        s = ch.ChannelSampler(probs, gold[f"c{ci}_transform"], seed=int(gold[f"c{ci}_seed"]))
        assert np.array_equal(s.signature_matrix, gold[f"c{ci}_signature_matrix"])
        assert len(s.channels) == int(gold[f"c{ci}_n_simplified"])
        for i, c in enumerate(s.channels):
            assert tuple(c.unique_col_ids) == tuple(gold[f"c{ci}_s{i}_ids"].tolist())
            assert np.array_equal(c.probs, gold[f"c{ci}_s{i}_probs"]), (ci, i)  # float64 bit-exact
        for tag, n in (("a", 257), ("b", 1), ("c", 4096)):
            assert np.array_equal(s.sample(n), gold[f"c{ci}_sample_{tag}"]), (ci, tag)


def test_x_error_detector_kat(gold):
    """test/integration/test_sampler_circuits.py:25-37: 4 of 10 shots fire (rows 1, 6, 8, 9)."""
    seed = int(np.random.default_rng(1).integers(0, 2**30))
    assert seed == int(gold["kat_channel_seed"]) == 508082495
    s = ch.ChannelSampler([ch.error_probs(0.3)], np.array([[1]], np.uint8), seed=seed)
    got = s.sample(10)
    assert np.array_equal(got, gold["kat_sample10"])
    assert np.flatnonzero(got[:, 0]).tolist() == [1, 6, 8, 9]


def test_xor_convolve_and_constructors():
    """unit/noise/test_channels.py: constructors sum to one; convolution of two 1-bit channels."""
    for p in (ch.error_probs(0.1), ch.pauli_channel_1_probs(0.01, 0.02, 0.03),
              ch.heralded_pauli_channel_1_probs(0.1, 0.01, 0.02, 0.03), ch.correlated_error_probs([0.1, 0.2, 0.3])):
        assert np.isclose(p.sum(), 1.0)
    c = ch.xor_convolve(ch.error_probs(0.1), ch.error_probs(0.2))
    assert np.allclose(c, [0.9 * 0.8 + 0.1 * 0.2, 0.1 * 0.8 + 0.9 * 0.2])
    with pytest.raises(ValueError):
        ch.Channel(np.array([0.5, 0.6]), (0,))
    with pytest.raises(ValueError):
        ch.xor_convolve(np.ones(2) / 2, np.ones(4) / 4)


def test_statistical_equivalence_after_simplification():
    """unit/noise/test_channels.py:212-235 in spirit: marginals of f match the unsimplified model."""
    rng = np.random.default_rng(0)
    probs = [ch.error_probs(0.05), ch.error_probs(0.1), ch.pauli_channel_1_probs(0.02, 0.03, 0.04)]
    T = np.array([[1, 1, 0, 1], [0, 1, 1, 0]], dtype=np.uint8)
    s = ch.ChannelSampler(probs, T, seed=3)
    f = s.sample(400_000)
    e = np.zeros((400_000, 4), np.uint8)
    e[:, 0] = rng.random(400_000) < 0.05
    e[:, 1] = rng.random(400_000) < 0.1
    o = rng.choice(4, size=400_000, p=probs[2])
    e[:, 2], e[:, 3] = o & 1, o >> 1
    want = (e @ T.T) % 2
    assert np.allclose(f.mean(axis=0), want.mean(axis=0), rtol=0.05)
    assert np.isclose((f[:, 0] & f[:, 1]).mean(), (want[:, 0] & want[:, 1]).mean(), rtol=0.08)


def test_sample_packed_is_the_same_stream_as_sample():
    """sample_packed == packbits(sample) for the same seed, and the two can be mixed mid-stream."""
    from tsim_amd.channels import ChannelSampler, error_probs, pauli_channel_1_probs

    rng = np.random.default_rng(3)
    for num_f, n_ch in [(5, 3), (64, 20), (70, 25), (130, 40)]:
        probs, cols = [], 0
        for k in range(n_ch):
            if k % 3 == 0:
                probs.append(pauli_channel_1_probs(0.01, 0.02, 0.03))
                cols += 2
            else:
                probs.append(error_probs(0.05 + 0.01 * (k % 4)))
                cols += 1
        T = rng.integers(0, 2, size=(num_f, cols), dtype=np.uint8)
        a = ChannelSampler(probs, T, seed=11)
        b = ChannelSampler(probs, T, seed=11)
        wf = (num_f + 63) // 64
        for n in (1, 777, 5000):
            want = a.sample(n)
            pad = np.zeros((n, wf * 64), np.uint8)
            pad[:, :num_f] = want
            want_packed = np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(n, wf)
            np.testing.assert_array_equal(b.sample_packed(n), want_packed)
        # interleave: packed on one, unpacked on the other, then swap
        x = a.sample_packed(300)
        y = b.sample(300)
        pad = np.zeros((300, wf * 64), np.uint8)
        pad[:, :num_f] = y
        np.testing.assert_array_equal(x, np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(300, wf))
        np.testing.assert_array_equal(a.sample(50), b.sample(50))
