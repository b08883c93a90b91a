"""Randomised parity: many small programs of diverse shape, all three kernels vs the C oracle."""

import numpy as np
import pytest

from oracle import oracle_c as OC
from tsim_amd import synth

pytestmark = pytest.mark.gpu


def random_program(rng):
    num_f = int(rng.integers(1, 131))
    n_comp = int(rng.integers(0, 5))
    comps = []
    for _ in range(n_comp):
        n = int(rng.integers(0, 6))
        F = int(rng.integers(0, min(num_f, rng.choice([8, 33, 45, 70])) + 1))
        kw = dict(
            ta=(0, int(rng.integers(0, 9))), tb=(0, int(rng.integers(0, 9))), tc=(0, int(rng.integers(0, 9))),
            td=(0, int(rng.integers(0, 4))), density=float(rng.choice([0.05, 0.3, 0.6])),
            zero_phase_fraction=float(rng.choice([0.0, 0.1, 0.5])),
        )
        comps.append(dict(n=n, F=F, G=[int(rng.integers(0, 7)) for _ in range(n + 1)], **kw))
    n_direct = int(rng.integers(0, min(num_f, 70) + 1))
    approx = bool(rng.integers(0, 2))
    comps = [dict(c, approx=approx) for c in comps]
    prog = synth.synth_program(
        num_f=num_f, n_direct=n_direct, components=comps, seed=int(rng.integers(0, 2**31)),
        shuffle_outputs=bool(rng.integers(0, 2)), direct_flip_fraction=0.3,
        identity_direct=bool(rng.integers(0, 2)),
    )
    return prog, num_f


def random_physical_program(rng):
    """Normalised probability models of random shape, narrow and wide (up to ~300 selected f bits): every kernel
    family gets programs it is eligible for (chunk tables, pattern tables, sparse columns, rows)."""
    num_f = int(rng.choice([8, 40, 64, 100, 200, 330, 500]))
    n_comp = int(rng.integers(1, 4))
    comps = []
    for _ in range(n_comp):
        n = int(rng.integers(1, 6))
        F = int(rng.integers(1, min(num_f, int(rng.choice([12, 40, 60, 120, 250]))) + 1))
        g0 = int(rng.integers(1, 4))
        G = [g0]
        for _k in range(n):
            G.append(G[-1] + int(rng.integers(0, G[-1] + 1)))
        comps.append(dict(n=n, F=F, G=G, density=float(rng.choice([0.05, 0.2])), ta=(0, 8), tb=(0, 8), tc=(0, 10), td=(0, 3)))
    n_direct = int(rng.integers(0, min(num_f, 140) + 1))
    prog = synth.physical_program(
        num_f=num_f, n_direct=n_direct, components=comps, seed=int(rng.integers(0, 2**31)),
        shuffle_outputs=bool(rng.integers(0, 2)), direct_flip_fraction=0.3, identity_direct=bool(rng.integers(0, 2)),
        approx=bool(rng.integers(0, 4) == 0),
    )
    return prog, num_f


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_physical_all_kernels(hip, seed):
    rng = np.random.default_rng(7000 + seed)
    prog, num_f = random_physical_program(rng)
    B = int(rng.choice([1, 64, 65, 257, 1000, 2500]))
    f = synth.synth_f(B, num_f, float(rng.choice([0.0, 0.01, 0.05, 0.3])), seed=seed)
    key = (int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)))
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, key, return_devs=True, return_overflow=True)
    assert not ov
    kinds = set()
    for mode in ("auto", "rows", "faithful"):
        hp = hip.HipProgram(prog, mode=mode)
        i = hp.info()
        kinds.add((i["chunk_table_kernel"], i["wide_sparse_kernel"], i["pattern_tables"]))
        got, gdev = hp.sample_batch(f, key)
        np.testing.assert_array_equal(got, want, err_msg=f"mode={mode} {i}")
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32), err_msg=f"mode={mode}")
        hp.close()


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_all_kernels(hip, seed):
    rng = np.random.default_rng(1000 + seed)
    prog, num_f = random_program(rng)
    B = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 2500]))
    f = synth.synth_f(B, num_f, float(rng.choice([0.0, 0.02, 0.3])), seed=seed)
    key = (int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)))
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, key, return_devs=True, return_overflow=True)
    if ov:
        pytest.skip("the reference's int32 arithmetic would wrap on this input")
    for mode in ("auto", "rows", "faithful"):
        hp = hip.HipProgram(prog, mode=mode)
        got, gdev = hp.sample_batch(f, key)
        np.testing.assert_array_equal(got, want, err_msg=f"mode={mode}")
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32), err_msg=f"mode={mode}")
        packed, _ = hp.sample_batch(f, key, bit_packed=True)
        if prog.num_outputs:
            ref = np.packbits(want, axis=1, bitorder="little")
            np.testing.assert_array_equal(packed[:, : ref.shape[1]], ref)


def test_zero_shots_and_zero_outputs(hip):
    prog, cfg = synth.config_program("C2")
    hp = hip.HipProgram(prog)
    out, _ = hp.sample_batch(np.zeros((0, cfg["num_f"]), np.uint8), (1, 2))
    assert out.shape == (0, 20) and out.dtype == np.bool_
    from tsim_amd.program import make_program

    empty = make_program([], [], 0, 0)
    assert hip.sample_program(empty, np.zeros((7, 3), np.uint8), (1, 2)).shape == (7, 0)
    with pytest.raises(ValueError):
        hp.sample_batch(np.zeros((4, 10), np.uint8), (1, 2))  # num_f smaller than the program needs
    with pytest.raises(ValueError):
        hp.sample_batch(np.zeros(5, np.uint8), (1, 2))
