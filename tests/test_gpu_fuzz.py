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


TUNE_CHOICES = {
    "defer_hard": [0, 1], "defer_group": [1, 3, 4, 8], "lw_fast": [0, 1], "wide_fused": [0, 1], "hard_wave": [0, 1],
    "hard_wave_rows": [0, 64, 1024, 100000], "hard_inline_rows": [0, 1 << 40], "hard_comp_par": [0, 1], "deep_after": [1, 20000000000],
    "fused_lanes": [0, 1, 2, 3], "fused_max": [1, 3, 8, 16], "wide": [0, 1], "wide_tables": [0, 1], "wide_compact": [0, 1], "hard_overflow": [0, 1],
}
PUBLIC_CHOICES = {"TSIM_AMD_ADAPTIVE": ["0", "1"], "TSIM_AMD_FUSED_STEPS": ["0", "1"], "TSIM_AMD_DEEP_TABLES": ["-1", "0", "1"]}


def random_switches(rng) -> dict:
    """A random assignment of the library's launch-plan switches: results must not depend on them."""
    env = {k: str(rng.choice(v)) for k, v in PUBLIC_CHOICES.items() if rng.random() < 0.5}
    tune = [f"{k}={rng.choice(v)}" for k, v in TUNE_CHOICES.items() if rng.random() < 0.5]
    if tune:
        env["TSIM_AMD_TUNE"] = ",".join(tune)
    return env


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_random_switch_assignments_through_the_steps_api(hip, seed, monkeypatch):
    """Random normalised programs (narrow, several components, wide) x a random switch assignment per handle: several batches
    through tsim_sample_steps_device - twice, so that the launch plan has feedback - and the serial API, against the C oracle."""
    from test_gpu_steps import _run_steps, _subkeys
    from tsim_amd import prng

    rng = np.random.default_rng(9100 + seed)
    prog, num_f = random_physical_program(rng)
    B, n = int(rng.choice([1, 65, 700, 3000])), int(rng.integers(1, 7))
    fs = [synth.synth_f(B, num_f, float(rng.choice([0.0, 0.01, 0.05, 0.3])), seed=50 * seed + i) for i in range(n)]
    key = prng.key(seed)
    _, subs = _subkeys(key, n)
    orc = OC.OracleProgram(prog)
    want = [np.packbits(orc.sample_program(f, k), axis=1, bitorder="little") for f, k in zip(fs, subs)]
    for _trial in range(2):
        env = random_switches(rng)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hp = hip.HipProgram(prog)
        for k in env:
            monkeypatch.delenv(k)
        packed = bool(rng.integers(0, 2))
        for _round in range(2):
            outs, _ = _run_steps(hp, prog, fs, key, num_f, packed=packed)
            for i in range(n):
                np.testing.assert_array_equal(outs[i], want[i], err_msg=f"switches {env}, batch {i}, round {_round}, {hp.info()}")
        got, _ = hp.sample_batch(fs[0], subs[0], bit_packed=True)
        np.testing.assert_array_equal(got[:, : want[0].shape[1]], want[0], err_msg=f"switches {env}, serial API")
        hp.close()
