"""Device-side noise sampler: distribution tests against the exact channel model and the host sampler."""

import numpy as np
import pytest

from tsim_amd import synth
from tsim_amd.channels import ChannelSampler, error_probs, pauli_channel_1_probs, correlated_error_probs

pytestmark = pytest.mark.gpu


def model():
    probs = [error_probs(0.01), error_probs(0.2), pauli_channel_1_probs(0.02, 0.03, 0.05),
             correlated_error_probs([0.1, 0.05, 0.02]), error_probs(0.001), error_probs(1.0)]
    rng = np.random.default_rng(1)
    T = (rng.random((70, 9)) < 0.2).astype(np.uint8)
    T[:, 8] = 0
    T[3, 8] = 1  # the always-firing channel flips f3
    return probs, T


def exact_marginals(probs, T, trials=2_000_000, seed=9):
    """Monte-Carlo of the UNsimplified model with numpy (independent of ChannelSampler)."""
    rng = np.random.default_rng(seed)
    e = []
    for p in probs:
        k = int(np.log2(len(p)))
        o = rng.choice(len(p), size=trials, p=p)
        e.extend([((o >> i) & 1).astype(np.uint8) for i in range(k)])
    e = np.stack(e, axis=1)
    f = (e @ T.T) % 2
    return f.mean(axis=0), (f[:, 0] & f[:, 1]).mean(), f


def test_device_noise_matches_model(hip):
    probs, T = model()
    prog, _ = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    cs = ChannelSampler(probs, T, seed=5)
    dn = hip.DeviceNoiseSampler(hp, cs)
    n = 1_000_000
    f = dn.sample(n, (11, 12))
    assert f.shape == (n, 70) and set(np.unique(f)) <= {0, 1}
    want, _, fm = exact_marginals(probs, T)
    got = f.mean(axis=0)
    sigma = np.sqrt(np.maximum(want * (1 - want), 1e-9) * (1 / n + 1 / len(fm)))
    assert np.all(np.abs(got - want) < 6 * sigma + 1e-6), np.max(np.abs(got - want) / sigma)
    # joint statistics of two f bits driven by shared channels
    for (i, j) in [(0, 1), (2, 5), (10, 40)]:
        a, b = (f[:, i] & f[:, j]).mean(), (fm[:, i] & fm[:, j]).mean()
        s = np.sqrt(max(b * (1 - b), 1e-9) * (1 / n + 1 / len(fm)))
        assert abs(a - b) < 6 * s + 1e-6
    # host sampler agrees too
    fh = cs.sample(500_000)
    assert np.all(np.abs(fh.mean(axis=0) - got) < 6 * np.sqrt(np.maximum(want * (1 - want), 1e-9) * (1 / n + 1 / 500_000)) + 1e-6)


def test_device_noise_deterministic_and_key_dependent(hip):
    probs, T = model()
    prog, _ = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    dn = hip.DeviceNoiseSampler(hp, ChannelSampler(probs, T, seed=5))
    a = dn.sample(50_000, (1, 2))
    b = dn.sample(50_000, (1, 2))
    c = dn.sample(50_000, (1, 3))
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert dn.sample(0, (1, 2)).shape == (0, 70)


def test_device_noise_no_channels_and_tiny_p(hip):
    prog, _ = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    empty = hip.DeviceNoiseSampler(hp, ChannelSampler([], np.zeros((5, 0), np.uint8), seed=1))
    assert not empty.sample(1000, (1, 1)).any()
    tiny = hip.DeviceNoiseSampler(hp, ChannelSampler([error_probs(1e-5)] * 3, np.eye(3, dtype=np.uint8), seed=1))
    f = tiny.sample(4_000_000, (3, 4))
    assert abs(f.mean() - 1e-5) < 6 * np.sqrt(1e-5 / (3 * 4_000_000))


def test_rare_outcomes_are_drawn(hip):
    """The reference draws a float64 uniform against a float64 CDF (src/tsim/noise/channels.py:641-656).  The device twin
    compares a 32-bit draw with the CDF rounded up to 2^-32 - round 3 compared a 24-bit uniform with a float32 CDF, and a
    conditional outcome below 6e-8 could never come out.  One channel: fires half the time; given that, outcome B has
    probability 4e-8 (the float32 CDF in front of it rounds to 1.0).  Over 1.2e9 shots B is expected 24 times."""
    probs = [np.array([0.5, 0.5 - 2e-8, 2e-8, 0.0])]
    T = np.eye(2, dtype=np.uint8)  # f0 = e0, f1 = e1: outcome A -> f = (1, 0), B -> (0, 1)
    prog, _ = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    cs = ChannelSampler(probs, T, seed=1)
    dn = hip.DeviceNoiseSampler(hp, cs)
    n, chunks = 20_000_000, 60
    buf = hp.malloc(n * 8)
    packed = np.zeros((n, 8), np.uint8)
    count = {0: 0, 1: 0, 2: 0, 3: 0}
    for c in range(chunks):
        dn.sample_into(buf.ptr, n, (77, c))
        hp.d2h(packed, buf)
        v = packed[:, 0]
        assert not packed[:, 1:].any()
        for k in count:
            count[k] += int(np.count_nonzero(v == k))
    buf.free()
    hp.close()
    total = n * chunks
    fires = count[1] + count[2] + count[3]
    assert abs(fires - 0.5 * total) < 6 * np.sqrt(0.25 * total)
    assert count[3] == 0
    want = 2e-8 * total  # 24
    assert 0 < count[2] and abs(count[2] - want) < 6 * np.sqrt(want) + 1, count
