import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def hip():
    """The HIP backend module; GPU tests fail (not skip) if it cannot be loaded."""
    from tsim_amd import _lib, backend

    _lib.load()
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible - `-m gpu` tests need an MI355X")
    return backend


def run_batches(sample_fn, program, seed, shots_list, num_f=0, f_list=None):
    """Drive ``sample_fn(program, f, subkey)`` the way _sample_batches does (sampler.py:392-400)."""
    from tsim_amd import prng

    k = prng.key(seed)
    outs = []
    for i, shots in enumerate(shots_list):
        k, sub = prng.split(k)
        f = np.zeros((shots, num_f), np.uint8) if f_list is None else f_list[i]
        outs.append(sample_fn(program, f, sub))
    return outs
