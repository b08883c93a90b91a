"""H4 matmul_gf2 on the HIP path at the reference's regression sizes
(/root/reference/test/unit/utils/test_linalg.py:102-126: all-ones rows with P in {256, 300, 1024} - the
float32 sum must not saturate at 255 - and random rows against an int64 matmul), on the W = 8 / 12 / 32
row kernels (and W = 48 / 64 beyond the reference's own sizes), plus sampled components with up to ~2000 parameters."""

import numpy as np
import pytest

from oracle import oracle_np as O
from tsim_amd import synth
from tsim_amd.program import CompiledComponent, empty_scalar_graphs, make_program, scalar_graphs_from_terms

pytestmark = pytest.mark.gpu


def _parity_program(P, rows):
    """One level whose graph g has amplitude w^(4 <rows[g], x>) = (-1)^parity (a HalfPi term of coefficient 4)."""
    lv = scalar_graphs_from_terms(P, [dict(B=[(4, r)]) for r in rows])
    comp = CompiledComponent(tuple(range(P)), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv))
    return make_program([comp], [], P, 0), lv


@pytest.mark.parametrize("mode", ["auto", "faithful"])
@pytest.mark.parametrize("P", [255, 256, 257, 300, 301, 512, 1023, 1024, 1025, 1536, 1537, 2047, 2048])
def test_all_ones_parity_does_not_saturate(hip, P, mode):
    prog, _ = _parity_program(P, [list(range(P))])
    hp = hip.HipProgram(prog, mode=mode)
    x = np.ones((3, P), np.uint8)
    x[1, 0] = 0  # one bit fewer: parity flips
    x[2, :] = 0
    z = hp.evaluate(0, 1, x)
    want = np.array([(-1.0) ** (P % 2), (-1.0) ** ((P - 1) % 2), 1.0], np.float32)
    np.testing.assert_array_equal(z.real, want)
    np.testing.assert_array_equal(z.imag, np.zeros(3, np.float32))


@pytest.mark.parametrize("P", [64, 300, 1024, 1300, 2048])
def test_random_rows_match_int64_matmul(hip, P):
    """test_linalg.py:88-101 shape: G graphs x 1 term; one graph per launch row via separate single-graph levels
    would be slow - instead the G parities are read off G one-graph levels of the same component."""
    rng = np.random.default_rng(P)
    G, B = 6, 64
    a = rng.integers(0, 2, size=(G, P), dtype=np.uint8)
    x = rng.integers(0, 2, size=(B, P), dtype=np.uint8)
    want = (x.astype(np.int64) @ a.astype(np.int64).T) % 2  # [B, G]
    for g in range(G):
        prog, lv = _parity_program(P, [np.flatnonzero(a[g]).tolist()])
        z = hip.HipProgram(prog).evaluate(0, 1, x)
        np.testing.assert_array_equal(z.real, np.where(want[:, g] == 1, -1.0, 1.0).astype(np.float32))
        np.testing.assert_array_equal(O.matmul_gf2(a[g][None, None, :], x)[:, 0, 0], want[:, g])


@pytest.mark.parametrize("mode", ["auto", "faithful"])
@pytest.mark.parametrize("F", [250, 600, 1000, 1400, 2040])
def test_wide_component_samples_match_oracle(hip, F, mode):
    """A sampled component with F + n = 253 .. 2043 parameters (W = 8 .. 64 words per row; the reference has no limit,
    utils/linalg.py:81-102 - this build stops at TSIM_MAX_PARAMS = 2048)."""
    prog = synth.synth_program(
        num_f=F + 20, n_direct=4,
        components=[dict(n=3, F=F, G=[2, 3, 3, 4], ta=(1, 5), tb=(1, 6), tc=(1, 6), td=(0, 2), density=0.04)], seed=F,
    )
    f = synth.synth_f(700, F + 20, 0.01, seed=F + 1)
    want, wdev = O.sample_program(prog, f, (3, F), return_devs=True)
    hp = hip.HipProgram(prog, mode=mode)
    got, gdev = hp.sample_batch(f, (3, F))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
    phys = synth.physical_program(num_f=F + 20, n_direct=4, components=[dict(n=3, F=F, G=[2, 3, 3, 4], density=0.04)], seed=F)
    want = O.sample_program(phys, f, (4, F))
    got, _ = hip.HipProgram(phys, mode=mode).sample_batch(f, (4, F))
    np.testing.assert_array_equal(got, want)
