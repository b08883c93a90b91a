"""End to end through CompiledDetectorSampler.sample() (the reference's host loop, src/tsim/sampler.py:340-420 + epilogue
:850-868) for program shapes the round-5 kernels opened - several wide-path components, 9-byte bit_packed rows, 65..128
parameters, more than 255 selected bits, 260 outputs: the default engine against the row-kernel engine (mode="rows") on the
same noise stream and key chain (bit-identical rows, every output format), and the device-noise pipeline reproducible and
statistically equal."""

import numpy as np
import pytest

from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler

pytestmark = pytest.mark.gpu

CLASSES = ["2wide", "narrow+wide", "F255", "F60", "F70", "F300", "out260_f320", "6narrow_f320",
           "n16", "n24", "20narrow"]  # (round 6: prefix-tree tables, more than 16 components)


def _noise(num_f, p):
    return dict(channel_probs=[error_probs(p)] * num_f, error_transform=np.eye(num_f, dtype=np.uint8))


@pytest.mark.parametrize("name", CLASSES)
def test_default_engine_equals_row_kernel_engine(hip, name):
    prog, c = synth.shape_class_program(name)
    noise = _noise(c["num_f"], c["p_bit"])
    n = 20_000
    a = CompiledDetectorSampler(prog, seed=9, **noise).sample(n, batch_size=6_000)
    b = CompiledDetectorSampler(prog, seed=9, mode="rows", **noise).sample(n, batch_size=6_000)
    assert a.shape == b.shape == (n, prog.num_detectors) and a.dtype == np.bool_
    np.testing.assert_array_equal(a, b)
    pa = CompiledDetectorSampler(prog, seed=9, **noise).sample(n, batch_size=6_000, bit_packed=True, append_observables=True)
    full = CompiledDetectorSampler(prog, seed=9, mode="rows", **noise).sample(n, batch_size=6_000, append_observables=True)
    np.testing.assert_array_equal(pa, np.packbits(full, axis=1, bitorder="little"))


@pytest.mark.parametrize("name", ["2wide", "F70", "F255", "n24"])
def test_device_noise_pipeline(hip, name):
    prog, c = synth.shape_class_program(name)
    noise = _noise(c["num_f"], c["p_bit"])
    n = 200_000
    a = CompiledDetectorSampler(prog, seed=4, noise="device", **noise).sample(n, batch_size=50_000, append_observables=True)
    a2 = CompiledDetectorSampler(prog, seed=4, noise="device", **noise).sample(n, batch_size=50_000, append_observables=True)
    np.testing.assert_array_equal(a, a2)
    b = CompiledDetectorSampler(prog, seed=4, noise="host", **noise).sample(n, batch_size=50_000, append_observables=True)
    pa, pb = a.mean(axis=0), b.mean(axis=0)
    sigma = np.sqrt(np.maximum(pb * (1 - pb), 1e-9) * 2 / n)
    assert np.all(np.abs(pa - pb) < 6 * sigma + 1e-6), (np.abs(pa - pb) / sigma).max()
