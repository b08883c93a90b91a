"""The native channel sampler (csrc/tsim_pcg.cpp) reproduces numpy's stream bit for bit.

* PCG64 raw outputs, uniform doubles, ziggurat exponentials and geometric draws, draw for draw, for several
  seeds and mid-stream states, with the generator left in the same state;
* the ziggurat's table boundaries: every acceptance threshold ke[idx] (ri = ke - 1 accepted, ri = ke not), the
  strip widths, the tail and wedge-test decisions right at the strip edges, by FORCING numpy's PCG64 to emit
  chosen values (scripts/numpy_ziggurat_tables.py) and comparing native and numpy on the same state;
* ChannelSampler: engine="native" == engine="numpy" on the golden cases (rows and generator state), so the
  golden vectors made by the reference module pin both.
"""

import ctypes as C
import os
import sys

import numpy as np
import pytest

from tsim_amd import _lib
from tsim_amd import channels as ch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from numpy_ziggurat_tables import Forced  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "channels_golden.npz")
RAW, DOUBLE, EXPO, GEOM = 0, 1, 2, 3


def native_draw(gen, kind, n, p=0.0):
    st, raw = ch._export_state(gen)
    out = np.empty(n, dtype={RAW: np.uint64, DOUBLE: np.float64, EXPO: np.float64, GEOM: np.int64}[kind])
    _lib.check(_lib.load().tsim_pcg_draw(C.byref(raw), kind, float(p), n, out.ctypes.data_as(C.c_void_p)), "tsim_pcg_draw")
    ch._import_state(gen, st, raw)
    return out


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**63 + 11])
def test_streams_equal_numpy_draw_for_draw(seed):
    a, b = np.random.default_rng(seed), np.random.default_rng(seed)
    np.testing.assert_array_equal(a.bit_generator.random_raw(1000), native_draw(b, RAW, 1000))
    for _ in range(3):  # interleave kinds: state hand-over in both directions
        np.testing.assert_array_equal(a.uniform(size=5000), native_draw(b, DOUBLE, 5000))
        np.testing.assert_array_equal(a.standard_exponential(300_000), native_draw(b, EXPO, 300_000))
        for p in (1e-9, 1e-4, 0.001, 0.02, 0.2, 0.3333, 0.34, 0.5, 0.97, 1.0):
            np.testing.assert_array_equal(a.geometric(p, 20_000), native_draw(b, GEOM, 20_000, p))
        assert a.bit_generator.state == b.bit_generator.state
        np.testing.assert_array_equal(a.random(7), b.random(7))  # numpy continues from the native state


def test_ziggurat_boundaries_with_forced_states():
    fz, tw = Forced(), Forced()

    def both(v1, v2=None):
        fz.arm(v1, v2)
        tw.arm(v1, v2)
        want = fz.gen.standard_exponential()
        got = native_draw(tw.gen, EXPO, 1)[0]
        assert want == got or (np.isnan(want) and np.isnan(got)), (hex(v1), v2, want, got)
        assert fz.bg.state == tw.bg.state, (hex(v1), v2)
        return fz.consumed()

    top = (1 << 53) - 1
    for idx in range(256):
        # acceptance threshold: find it with numpy (binary search on the number of raw values consumed) ...
        lo, hi = -1, 1 << 53
        while hi - lo > 1:
            mid = (lo + hi) // 2
            fz.arm((mid << 11) | (idx << 3))
            fz.gen.standard_exponential()
            if fz.consumed() == 1:
                lo = mid
            else:
                hi = mid
        # ... and require the native sampler to agree on both sides of it, at the ends and in the unlikely branch
        for ri in {max(lo, 0), hi if hi <= top else top, 0, top, 1 << 52, (1 << 52) + 12345}:
            for u in (0, 1, 2, (1 << 52), top - 1, top):
                both((ri << 11) | (idx << 3), u << 11)
        # wedge decisions right below the strip edge, u swept around the flip
        for back in (1, 2, 3, 9, 100):
            ri = top + 1 - back
            if ri <= lo:
                continue
            a, b = 0, top
            while b - a > 1:
                mid = (a + b) // 2
                fz.arm((ri << 11) | (idx << 3), mid << 11)
                fz.gen.standard_exponential()
                if fz.consumed() == 2:
                    a = mid
                else:
                    b = mid
            for u in range(max(0, a - 3), min(top, b + 3) + 1):
                both((ri << 11) | (idx << 3), u << 11)


def test_channel_sampler_native_equals_numpy_engine():
    ok, why = ch.native_stream_available()
    assert ok, why
    gold = np.load(GOLD)
    for ci in range(int(gold["n_cases"])):
        probs = [gold[f"c{ci}_probs{i}"] for i in range(int(gold[f"c{ci}_n_channels"]))]
        T, seed = gold[f"c{ci}_transform"], int(gold[f"c{ci}_seed"])
        nat = ch.ChannelSampler(probs, T, seed=seed, engine="native")
        ref = ch.ChannelSampler(probs, T, seed=seed, engine="numpy")
        assert nat._native is not None and ref._native is None
        for tag, n in (("a", 257), ("b", 1), ("c", 4096)):
            got = nat.sample(n)
            np.testing.assert_array_equal(got, gold[f"c{ci}_sample_{tag}"])  # the reference module's own output
            np.testing.assert_array_equal(got, ref.sample(n))
        for n in (100_000, 3, 65_537):
            np.testing.assert_array_equal(nat.sample_packed(n), ref.sample_packed(n))
            assert nat._rng.bit_generator.state == ref._rng.bit_generator.state


def test_native_big_batch_many_channels_threads():
    """64 one-bit channels at p = 0.02, 10^6 rows (the e2e benchmark's noise model): tiles + threads, equal to numpy."""
    probs = [ch.error_probs(0.02)] * 64 + [ch.pauli_channel_1_probs(0.01, 0.005, 0.002)] * 8
    T = np.concatenate([np.eye(64, dtype=np.uint8), np.random.default_rng(0).integers(0, 2, size=(64, 16), dtype=np.uint8)], axis=1)
    a = ch.ChannelSampler(probs, T, seed=5, engine="native")
    b = ch.ChannelSampler(probs, T, seed=5, engine="numpy")
    np.testing.assert_array_equal(a.sample_packed(1_000_000), b.sample_packed(1_000_000))
    np.testing.assert_array_equal(a.sample_packed(10), b.sample_packed(10))
    wide = ch.ChannelSampler([ch.error_probs(0.05)] * 200, np.eye(200, dtype=np.uint8), seed=6, engine="native")
    wide_np = ch.ChannelSampler([ch.error_probs(0.05)] * 200, np.eye(200, dtype=np.uint8), seed=6, engine="numpy")
    np.testing.assert_array_equal(wide.sample_packed(50_000), wide_np.sample_packed(50_000))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_blocked_stream_equals_scalar_and_numpy(monkeypatch, seed):
    """The AVX-512 path (four interleaved LCG chains into a buffer, eight geometric draws per step, positions by an
    in-register prefix sum, output skipping by count) against the scalar path of the same library (TSIM_PCG_SCALAR=1)
    and numpy: channel mixes that hit every branch - p from 1e-9 to 0.9 (tiny-p scalar fallback, inversion, search),
    one-outcome channels with more fires than the buffer holds (skip by re-seeding), multi-outcome channels (uniforms
    through the buffer), batches of 1 to 300 000 rows, consecutive calls on one generator."""
    rng = np.random.default_rng(100 + seed)
    probs = []
    for _ in range(int(rng.integers(3, 40))):
        p = float(rng.choice([1e-9, 1e-5, 1e-3, 0.02, 0.1, 0.3, 0.34, 0.5, 0.9]))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            probs.append(np.array([1.0 - p, p]))
        elif kind == 1:
            probs.append(ch.pauli_channel_1_probs(p / 3, p / 3, p / 3))
        else:
            w = rng.random(15)
            probs.append(np.concatenate([[1.0 - p], p * w / w.sum()]))
    nbits = sum(int(np.log2(len(q))) for q in probs)
    num_f = int(rng.integers(8, 130))
    T = rng.integers(0, 2, size=(num_f, nbits), dtype=np.uint8)  # [f bit, channel bit]
    T[rng.integers(0, num_f, size=nbits), np.arange(nbits)] = 1  # no empty column
    a = ch.ChannelSampler(probs, T, seed=seed, engine="native")
    b = ch.ChannelSampler(probs, T, seed=seed, engine="native")
    c = ch.ChannelSampler(probs, T, seed=seed, engine="numpy")
    assert a._native is not None and b._native is not None
    for n in (1, 7, 300_000, 64, 12_345):
        monkeypatch.delenv("TSIM_PCG_SCALAR", raising=False)
        fast = a.sample_packed(n).copy()
        monkeypatch.setenv("TSIM_PCG_SCALAR", "1")
        slow = b.sample_packed(n).copy()
        monkeypatch.delenv("TSIM_PCG_SCALAR", raising=False)
        np.testing.assert_array_equal(fast, slow)
        np.testing.assert_array_equal(fast, c.sample_packed(n))
        assert a._rng.bit_generator.state == b._rng.bit_generator.state == c._rng.bit_generator.state


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_prefetched_stream_equals_serial_stream_and_numpy(monkeypatch, seed):
    """Worker threads fill the raw outputs, the exponentials of draws starting at every output and their lengths ahead
    of the consumer (tsim_pcg.cpp: PStream); the consumer walks the channels over those arrays.  Against the serial
    blocked stream of the same library (TSIM_PCG_SERIAL=1) and numpy: channel mixes with one-outcome channels (skips),
    multi-outcome channels (uniforms from the prefetched outputs), p >= 1/3 (search), tiny p (saturating gaps), several
    thread counts, consecutive calls on one generator, rows and final generator state."""
    rng = np.random.default_rng(900 + seed)
    probs = [np.array([0.98, 0.02])] * 20
    for _ in range(12):
        p = float(rng.choice([1e-9, 1e-4, 0.02, 0.2, 0.34, 0.6]))
        w = rng.random(3)
        probs.append(np.concatenate([[1.0 - p], p * w / w.sum()]) if rng.integers(0, 2) else np.array([1.0 - p, p]))
    nbits = sum(int(np.log2(len(q))) for q in probs)
    T = rng.integers(0, 2, size=(70, nbits), dtype=np.uint8)
    T[rng.integers(0, 70, size=nbits), np.arange(nbits)] = 1
    ref = ch.ChannelSampler(probs, T, seed=40 + seed, engine="numpy")
    want = [ref.sample_packed(n) for n in (400_000, 150_000, 7)]
    for env in ({}, {"TSIM_PCG_THREADS": "3"}, {"TSIM_PCG_THREADS": "16"}, {"TSIM_PCG_SERIAL": "1"}):
        for k in ("TSIM_PCG_THREADS", "TSIM_PCG_SERIAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        nat = ch.ChannelSampler(probs, T, seed=40 + seed, engine="native")
        for n, w in zip((400_000, 150_000, 7), want):
            np.testing.assert_array_equal(nat.sample_packed(n), w, err_msg=str(env))
        assert nat._rng.bit_generator.state == ref._rng.bit_generator.state, env
