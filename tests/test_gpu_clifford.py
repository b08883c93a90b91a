"""Clifford front-end outputs that are NOT single f bits (XORs of several basis bits) run as
one-output components on the GPU; the bits must satisfy the same linear relations."""

import warnings

import numpy as np
import pytest

from tsim_amd.clifford import CliffordCircuit

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("noise", ["host", "device"])
def test_dependent_and_constant_detectors(hip, noise):
    c = CliffordCircuit("""
        R 0 1 2
        X 2
        X_ERROR(0.2) 0
        X_ERROR(0.3) 1
        M 0 1 2
        DETECTOR rec[-3]
        DETECTOR rec[-2]
        DETECTOR rec[-3] rec[-2]
        DETECTOR rec[-1]
        DETECTOR rec[-3] rec[-2] rec[-1]
        OBSERVABLE_INCLUDE(1) rec[-3]
    """)
    program, probs, et = c.compile()
    assert program.num_detectors == 5 and program.num_outputs == 6
    # d0, d1 are basis rows; d2 = d0^d1 and d4 = d0^d1^1 are components; d3 is the constant 1; obs = d0 again
    assert len(program.components) == 2 and et.shape == (3, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = c.compile_detector_sampler(seed=11, noise=noise)
        d = s.sample(40000, batch_size=16384, append_observables=True)
    assert d.shape == (40000, 6)
    np.testing.assert_array_equal(d[:, 2], d[:, 0] ^ d[:, 1])
    assert d[:, 3].all()
    np.testing.assert_array_equal(d[:, 4], ~(d[:, 0] ^ d[:, 1]))
    np.testing.assert_array_equal(d[:, 5], d[:, 0])
    assert abs(d[:, 0].mean() - 0.2) < 0.01 and abs(d[:, 1].mean() - 0.3) < 0.01


def test_host_and_oracle_agree_on_component_program(hip):
    """The hand-built delta components go through the same parity machinery as every other program."""
    from oracle import oracle_c as OC
    from tsim_amd import prng

    c = CliffordCircuit("R 0 1\nX_ERROR(0.5) 0 1\nM 0 1\nDETECTOR rec[-1]\nDETECTOR rec[-2]\nDETECTOR rec[-1] rec[-2]")
    program, _, et = c.compile()
    f = np.random.default_rng(0).integers(0, 2, size=(500, et.shape[0]), dtype=np.uint8)
    key = prng.key(3)
    want, wdev, ov = OC.OracleProgram(program).sample_program(f, key, return_devs=True, return_overflow=True)
    got, gdev = hip.HipProgram(program).sample_batch(f, key)
    assert not ov
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got[:, 2], got[:, 0] ^ got[:, 1])


def test_measurement_sampler_kats_from_circuit_text(hip):
    """The reference's seeded measurement-sampler counts, from circuit text through the GPU:
    test/unit/test_sampler.py:223-233 and test/integration/test_sampler_circuits.py:10-22,90-109."""
    s = CliffordCircuit("H 0\nM 0").compile_sampler(seed=0)
    assert [int(s.sample(100).sum()) for _ in range(4)] == [48, 53, 52, 50]
    m = CliffordCircuit("R 0 1\nH 0\nCNOT 0 1\nM 0 1").compile_sampler(seed=0).sample(100)
    assert np.array_equal(m[:, 0], m[:, 1]) and int(m[:, 0].sum()) == 48
    m = CliffordCircuit("RX 0\nRX 0\nM 0\nRX 0\nM 0\nR 0\nM 0").compile_sampler(seed=0).sample(10)
    assert m.sum(axis=0).tolist() == [7, 4, 0]


def test_measurement_sampler_of_a_noisy_ghz_state(hip):
    c = CliffordCircuit("R 0 1 2 3\nH 0\nCX 0 1 1 2 2 3\nX_ERROR(0.25) 3\nM 0 1 2 3")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = c.compile_sampler(seed=2).sample(20000, batch_size=8192)
    assert np.array_equal(m[:, 0], m[:, 1]) and np.array_equal(m[:, 1], m[:, 2])
    assert abs(m[:, 0].mean() - 0.5) < 0.02
    assert abs((m[:, 3] ^ m[:, 0]).mean() - 0.25) < 0.02


def test_teleportation_with_feedback_and_new_gates(hip):
    """Classically controlled corrections: the two Bell-measurement outcomes are fair coins, the
    teleported state reads out deterministically; a record error (1/8) applies the wrong correction."""
    text = """
        R 0 1 2
        C_XYZ 0          # |0> -> |+>  (Z -> X)
        H 1
        CX 1 2
        CX 0 1
        H 0
        M(0.125) 0 1
        CX rec[-1] 2
        CZ rec[-2] 2
        MX 2
    """
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = CliffordCircuit(text).compile_sampler(seed=4).sample(40000, batch_size=16384)
    assert abs(m[:, 0].mean() - 0.5) < 0.02 and abs(m[:, 1].mean() - 0.5) < 0.02
    assert abs((m[:, 0] ^ m[:, 1]).mean() - 0.5) < 0.02
    assert abs(m[:, 2].mean() - 0.125) < 0.01          # the Z correction's record error flips the X readout
    m = CliffordCircuit(text.replace("M(0.125)", "M").replace("MX 2", "ISWAP 2 3\nSQRT_YY 3 4\nSQRT_YY 3 4\nMY 3\nM 4")
                        ).compile_sampler(seed=4).sample(4096)
    # ISWAP sends the stabilizers X2, Z3 to Z2 Y3, Z2: qubit 3 holds +Y; SQRT_YY twice = YY flips qubit 4
    assert (m[:, 2] == 0).all() and (m[:, 3] == 1).all()


@pytest.mark.parametrize("d", [3, 5, 7])
def test_surface_code_memory_on_the_device_equals_the_host_path(hip, d):
    """Programs without compiled components (every output direct): large requests run on the GPU - one ChannelSampler call
    for all shots, as the reference's _sample_direct makes it (sampler.py:547-555), the streaming kernel for direct outputs
    (d = 3, 5: at most 128 f bits and outputs) or the row kernel (d = 7) through tsim_sample_steps_device.  Same bits as
    the host's numpy path for the same seed - bools, bit_packed, detectors only and with observables; with noise="device"
    the statistics agree."""
    from tsim_amd.circuits import rotated_surface_code_memory

    c = CliffordCircuit(rotated_surface_code_memory(d, 3, after_clifford_depolarization=2e-3, before_measure_flip_probability=1e-3))
    n = 70_000
    dev = c.compile_detector_sampler(seed=4)
    assert not dev._program.components and dev._direct_on_device(n)
    got = dev.sample(n, append_observables=True)
    host = c.compile_detector_sampler(seed=4)
    want = host._sample_direct(n)
    np.testing.assert_array_equal(got, want)
    nd = dev._num_detectors
    np.testing.assert_array_equal(c.compile_detector_sampler(seed=4).sample(n, bit_packed=True), np.packbits(want[:, :nd], axis=1, bitorder="little"))
    np.testing.assert_array_equal(c.compile_detector_sampler(seed=4).sample(n, bit_packed=True, append_observables=True),
                                  np.packbits(want, axis=1, bitorder="little"))
    dn = c.compile_detector_sampler(seed=4, noise="device").sample(200_000, append_observables=True)
    big = c.compile_detector_sampler(seed=9)._sample_direct(200_000)
    pa, pb = dn.mean(axis=0), big.mean(axis=0)
    sig = np.sqrt(np.maximum(pb * (1 - pb), 1e-9) * 2 / 200_000)
    assert np.all(np.abs(pa - pb) < 6 * sig + 1e-5), float(np.max(np.abs(pa - pb) / (sig + 1e-9)))
