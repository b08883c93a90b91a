"""GPU parity: the HIP path (through the C ABI) against the numpy oracle and the reference KATs."""

import warnings

import numpy as np
import pytest

from conftest import run_batches
from oracle import oracle_np as O
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


def hip_sample(hip):
    def fn(program, f, key):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            hp = hip.get_hip_program(program)
            out, _ = hp.sample_batch(f, key)
        return out

    return fn


# ---- the reference's seeded known-answer tests, through the HIP kernels ----


def test_kat_seed_counts(hip):
    """test/unit/test_sampler.py:223-233: H 0; M 0, seed 0 -> 48, 53, 52, 50."""
    outs = run_batches(hip_sample(hip), synth.kat_h_m(), 0, [100] * 4)
    assert [int(o.sum()) for o in outs] == [48, 53, 52, 50]


def test_kat_t_gate(hip):
    """test/integration/test_sampler_circuits.py:40-49 -> 9 of 100."""
    (o,) = run_batches(hip_sample(hip), synth.kat_t_gate(), 0, [100])
    assert int(o.sum()) == 9


def test_kat_r_gate(hip):
    """test_sampler_circuits.py:90-105 -> 7, 4, 0 of 10 (key threading across components)."""
    (o,) = run_batches(hip_sample(hip), synth.kat_r_gate(), 0, [10])
    assert o.sum(axis=0).tolist() == [7, 4, 0]


def test_kat_bell(hip):
    """test_sampler_circuits.py:10-22 -> 48 of 100, both bits equal."""
    (o,) = run_batches(hip_sample(hip), synth.kat_bell(), 0, [100])
    assert np.array_equal(o[:, 0], o[:, 1]) and int(o[:, 0].sum()) == 48


# ---- HIP vs oracle on seeded synthetic programs -----------------------------

SMALL = dict(
    num_f=40,
    n_direct=6,
    components=[
        dict(n=1, F=5, G=[2, 3], ta=(0, 4), tb=(0, 5), tc=(0, 6), td=(0, 2)),
        dict(n=3, F=12, G=[3, 4, 5, 6], ta=(0, 6), tb=(0, 8), tc=(0, 8), td=(0, 3)),
    ],
)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
@pytest.mark.parametrize("approx", [False, True])
def test_sample_matches_oracle_small(hip, seed, approx):
    comps = [dict(c, approx=approx) for c in SMALL["components"]]
    prog = synth.synth_program(
        num_f=SMALL["num_f"], n_direct=SMALL["n_direct"], components=comps, seed=seed,
        shuffle_outputs=True, direct_flip_fraction=0.3, identity_direct=False,
    )
    f = synth.synth_f(777, SMALL["num_f"], 0.1, seed=seed)
    key = (123 + seed, 456)
    want, wdev = O.sample_program(prog, f, key, return_devs=True)
    hp = hip.get_hip_program(prog)
    got, gdev = hp.sample_batch(f, key)
    assert got.shape == want.shape and got.dtype == np.bool_
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))


@pytest.mark.parametrize("approx", [False, True])
def test_evaluate_matches_oracle_exact_ints(hip, approx):
    rng = np.random.default_rng(5)
    lv = synth.synth_level(rng, 37, 12, approx=approx)
    from tsim_amd.program import CompiledComponent, empty_scalar_graphs, make_program

    comp = CompiledComponent(tuple(range(37)), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv))
    prog = make_program([comp], [], 37, 0)
    hp = hip.HipProgram(prog)
    pv = (rng.random((500, 37)) < 0.4).astype(np.uint8)
    z, ex = hp.evaluate(0, 1, pv, exact=True)
    want = O.evaluate(lv, pv)
    np.testing.assert_array_equal(z.view(np.float32), want.view(np.float32))  # bit-exact floats
    if not approx:
        coeffs, power = O.evaluate_exact(lv, pv)
        np.testing.assert_array_equal(ex[:, :4], coeffs)
        nz = np.any(coeffs != 0, axis=1)
        np.testing.assert_array_equal(ex[nz, 4], power[nz])


@pytest.mark.parametrize("physical", [True, False])
def test_c2_shape_program_matches_oracle(hip, physical):
    """BASELINE config C2 (35-qubit distillation shape), 2000 shots, against the oracle."""
    prog, cfg = synth.config_program("C2", physical=physical)
    f = synth.synth_f(2000, cfg["num_f"], cfg["p_bit"], seed=cfg["seed"])
    key = (7, 9)
    want = O.sample_program(prog, f, key)
    got, _ = hip.get_hip_program(prog).sample_batch(f, key)
    np.testing.assert_array_equal(got, want)


def test_shard_invariance(hip):
    """Sharding a batch over launches (shot_offset) reproduces the unsharded bits."""
    prog, cfg = synth.config_program("C2")
    f = synth.synth_f(3000, cfg["num_f"], cfg["p_bit"], seed=1)
    hp = hip.get_hip_program(prog)
    key = (11, 13)
    full, _ = hp.sample_batch(f, key)
    parts = [hp.sample_batch(f[a:b], key, shot_offset=a)[0] for a, b in [(0, 1000), (1000, 1001), (1001, 3000)]]
    np.testing.assert_array_equal(np.concatenate(parts), full)


def test_bit_packed_output(hip):
    prog, cfg = synth.config_program("C2")
    f = synth.synth_f(500, cfg["num_f"], cfg["p_bit"], seed=2)
    hp = hip.get_hip_program(prog)
    a, _ = hp.sample_batch(f, (1, 2))
    b, _ = hp.sample_batch(f, (1, 2), bit_packed=True)
    want = np.packbits(a, axis=1, bitorder="little")
    np.testing.assert_array_equal(b[:, : want.shape[1]], want)


def test_multi_device_sharding_same_bits(hip):
    """tsim_amd.dist.sample_program_multi_device over [0, 0, 0] == one launch (device-side pack path)."""
    from tsim_amd import dist as tdist

    prog, cfg = synth.config_program("C2")
    f = synth.synth_f(5003, cfg["num_f"], cfg["p_bit"], seed=4)
    key = (31, 37)
    full, _ = hip.get_hip_program(prog).sample_batch(f, key)
    got = tdist.sample_program_multi_device(prog, f, key, [0, 0, 0])
    np.testing.assert_array_equal(got, full)


def test_pack_unpack_kernels(hip):
    """k_pack_bits / k_unpack_bits against numpy.packbits for ragged widths."""
    prog, _ = synth.config_program("C2")
    hp = hip.get_hip_program(prog)
    rng = np.random.default_rng(0)
    for nbits in (1, 7, 16, 20, 63, 64, 65, 128, 200, 320):
        a = (rng.random((333, nbits)) < 0.5).astype(np.uint8) * rng.integers(1, 255, size=(333, nbits), dtype=np.uint8)
        wq = (nbits + 63) // 64
        d_in, d_p, d_out = hp.malloc(a.nbytes), hp.malloc(333 * wq * 8), hp.malloc(a.nbytes)
        hp.h2d(d_in, a)
        hp.pack_bits_device(d_in.ptr, 333, nbits, d_p.ptr)
        packed = np.zeros((333, wq * 8), np.uint8)
        hp.d2h(packed, d_p)
        want = np.packbits(a != 0, axis=1, bitorder="little")
        np.testing.assert_array_equal(packed[:, : want.shape[1]], want)
        assert not packed[:, want.shape[1]:].any()
        hp.unpack_bits_device(d_p.ptr, 333, nbits, d_out.ptr)
        back = np.zeros_like(a)
        hp.d2h(back, d_out)
        np.testing.assert_array_equal(back, (a != 0).astype(np.uint8))


# ---- both evaluation formulations against the oracle ---------------------------


@pytest.mark.parametrize("mode", ["auto", "rows", "faithful"])
@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5"])
def test_modes_match_oracle_on_baseline_shapes(hip, mode, name):
    """The unconstrained random programs (nonsense marginals: thresholds outside [0, 1], NaN): the kernels
    must equal the oracle there too.  The normalised ones are in test_gpu_physical.py."""
    from oracle import oracle_c as OC

    prog, cfg = synth.config_program(name, physical=False)
    n = 600 if name == "C4" else 3000
    f = synth.synth_f(n, cfg["num_f"], cfg["p_bit"] * 2, seed=17)
    hp = hip.HipProgram(prog, mode=mode)
    assert hp.fast == (mode != "faithful")
    assert hp.info()["chunk_table_kernel"] == (mode == "auto" and name != "C5")
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, (41, 43), return_devs=True, return_overflow=True)
    assert not ov
    got, gdev = hp.sample_batch(f, (41, 43))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))


@pytest.mark.parametrize("mode", ["auto", "rows", "faithful"])
@pytest.mark.parametrize("approx", [False, True])
def test_modes_evaluate_exact(hip, mode, approx):
    from oracle import oracle_c as OC
    from tsim_amd.program import CompiledComponent, empty_scalar_graphs, make_program

    rng = np.random.default_rng(23)
    lv = synth.synth_level(rng, 45, 20, approx=approx, zero_phase_fraction=0.3)
    comp = CompiledComponent(tuple(range(45)), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv))
    prog = make_program([comp], [], 45, 0)
    pv = (rng.random((2000, 45)) < 0.5).astype(np.uint8)
    z, ex = hip.HipProgram(prog, mode=mode).evaluate(0, 1, pv, exact=True)
    wz, wex, ov = OC.OracleProgram(prog).evaluate(0, 1, pv, exact=True)
    assert not ov
    np.testing.assert_array_equal(z.view(np.float32), wz.view(np.float32))
    if not approx:
        np.testing.assert_array_equal(ex[:, :4], wex[:, :4])
        nz = np.any(wex[:, :4] != 0, axis=1)
        np.testing.assert_array_equal(ex[nz, 4], wex[nz, 4])


def test_auto_falls_back_to_faithful_when_not_eligible(hip):
    """So many NodePhases terms in one graph that the reference's int32 scan may wrap -> faithful layout.  The bound (round 6,
    tsim_pack.hip: level_fast_eligible) prices a term by what it can add to log2 of the running coefficients under the scan's
    one-reduction-per-product rule: 0 for deltas (phase 0 / 4), 1/2 for phase 2 / 6, 0.886 for odd phases; 29 bits is the limit."""
    from tsim_amd.program import CompiledComponent, empty_scalar_graphs, make_program, scalar_graphs_from_terms

    pv = np.array([[0, 0, 0], [1, 0, 1], [1, 1, 1], [0, 1, 0]], np.uint8)
    cases = [
        ([(1 + 2 * (t % 4), [t % 3]) for t in range(40)], False),             # 40 odd phases: 36 bits
        ([(1 + 2 * (t % 4), [t % 3]) for t in range(31)], True),              # 31 odd phases: 28.5 bits - cannot wrap
        ([(4 * (t % 2), [t % 3, (t + 1) % 3][: 1 + t % 2]) for t in range(45)], True),   # 45 deltas: the running value never grows
        ([(2 + 4 * (t % 2), [t % 3]) for t in range(50)] + [(1, [0]), (7, [1])], True),  # 50 x (1 +- i) and two odd: 27.8 bits
        ([(2 + 4 * (t % 2), [t % 3]) for t in range(60)], False),             # 60 x (1 +- i): 31 bits
    ]
    for terms, want_fast in cases:
        lv = scalar_graphs_from_terms(3, [dict(A=terms)])
        comp = CompiledComponent((0, 1, 2), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv))
        prog = make_program([comp], [], 3, 0)
        hp = hip.HipProgram(prog)
        assert hp.fast == want_fast, (len(terms), hp.fast)
        z, ex = hp.evaluate(0, 1, pv, exact=True)
        coeffs, power = O.evaluate_exact(lv, pv)
        np.testing.assert_array_equal(ex[:, :4], coeffs)
        nz = np.any(np.asarray(coeffs) != 0, axis=1)
        np.testing.assert_array_equal(ex[nz, 4], np.asarray(power)[nz])
        hp.close()


@pytest.mark.parametrize("mode", ["auto", "rows", "faithful"])
@pytest.mark.parametrize("approx", [False, True])
def test_modes_small_programs_with_norm_check(hip, mode, approx):
    """Small multi-component programs (incl. shuffled outputs): samples + normalisation deviation."""
    comps = [dict(c, approx=approx) for c in SMALL["components"]]
    for seed in (5, 6):
        prog = synth.synth_program(
            num_f=SMALL["num_f"], n_direct=SMALL["n_direct"], components=comps, seed=seed,
            shuffle_outputs=True, direct_flip_fraction=0.3, identity_direct=False,
        )
        f = synth.synth_f(1500, SMALL["num_f"], 0.15, seed=seed)
        want, wdev = O.sample_program(prog, f, (9, seed), return_devs=True)
        got, gdev = hip.HipProgram(prog, mode=mode).sample_batch(f, (9, seed))
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))


def _delta_heavy_program():
    """Fuzz program 7048 of scripts/fuzz_steps.py (kind 8): components of 13 and 24 outputs, the second with 36 NodePhases terms per
    graph, most of them deltas.  At p >= 0.02 the REFERENCE's running sum of some levels wraps int32 (a zero-valued graph brings a
    low power, the carry is aligned to it: exact_scalar.py:74-84) - oracle.c flags it."""
    rng = np.random.default_rng(31000 + 7048)
    nf = int(rng.choice([64, 96, 160]))
    comps = []
    for _ in range(int(rng.integers(1, 3))):
        n = int(rng.integers(9, 25)); F = int(rng.integers(4, min(nf, 40)))
        G = [int(rng.integers(1, 4))]
        for _k in range(n):
            G.append(G[-1] + int(rng.integers(0, 3)))
        comps.append(dict(n=n, F=F, G=G, density=0.2, shared_delta=(0.8 if n > 16 else float(rng.choice([0.0, 0.5])))))
    prog = synth.physical_program(num_f=nf, n_direct=int(rng.integers(0, 30)), components=comps, seed=int(rng.integers(0, 2**31)),
                                  shuffle_outputs=bool(rng.integers(0, 2)), direct_flip_fraction=0.3, identity_direct=bool(rng.integers(0, 2)))
    return prog, nf


def test_where_the_reference_wraps_int32(hip):
    """The policy for the reference's own int32 overflow.  Products: ruled out statically, else the faithful formulation
    (test_auto_falls_back_to_faithful_when_not_eligible).  Running sums: ``info()["reference_sum_wrap_possible"]`` says whether the
    packer could rule a wrap out (the BASELINE configurations: yes); where it could not, the exact formulation still runs - it
    equals the reference on every input the reference's arithmetic does not wrap on, and ``mode="faithful"`` mirrors the wrap
    bit for bit."""
    from oracle import oracle_c as OC

    prog, nf = _delta_heavy_program()
    op = OC.OracleProgram(prog)
    auto = hip.HipProgram(prog)
    faithful = hip.HipProgram(prog, mode="faithful")
    assert auto.fast and auto.info()["reference_sum_wrap_possible"]
    assert not faithful.fast
    seen_wrap = False
    for p in (0.0, 0.02, 0.3):
        f = synth.synth_f(1000, nf, p, seed=3)
        want, wdev, ov = op.sample_program(f, prng.key(5), return_devs=True, return_overflow=True)
        got_f, dev_f = faithful.sample_batch(f, prng.key(5))
        np.testing.assert_array_equal(got_f, want, err_msg=f"faithful, p {p}, reference wraps: {bool(ov)}")
        np.testing.assert_array_equal(np.asarray(dev_f, np.float32), np.asarray(wdev, np.float32))
        got_a, _ = auto.sample_batch(f, prng.key(5))
        if ov:
            seen_wrap = True
            assert int((np.asarray(got_a) != want).any(axis=1).sum()) < 50  # (a handful of rows: those whose sums wrapped)
        else:
            np.testing.assert_array_equal(got_a, want, err_msg=f"auto, p {p}")
    assert seen_wrap
    auto.close(); faithful.close()
    import os
    os.environ["TSIM_AMD_MODE"] = "strict"  # the exact formulation only where no int32 operation of the reference can wrap
    try:
        strict = hip.HipProgram(prog)
        assert not strict.fast
        f = synth.synth_f(1000, nf, 0.3, seed=3)
        got_s, _ = strict.sample_batch(f, prng.key(5))
        np.testing.assert_array_equal(got_s, op.sample_program(f, prng.key(5)))
        strict.close()
        c2 = hip.HipProgram(synth.config_program("C2")[0])
        assert c2.fast
        c2.close()
    finally:
        os.environ.pop("TSIM_AMD_MODE", None)
    for name in ("C2", "C3", "C4", "C5"):
        hp = hip.HipProgram(synth.config_program(name)[0])
        assert hp.fast and not hp.info()["reference_sum_wrap_possible"], name
        hp.close()
