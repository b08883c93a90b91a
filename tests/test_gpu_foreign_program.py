"""The exporter / drop-in seam on the GPU with a duck-typed foreign program (no tsim here): a frozen,
non-numpy CompiledProgram goes through from_tsim -> HipProgram and through install() on a fake tsim package,
and config C1 (all-direct, 1000 shots) runs through backend.sample_program."""

import warnings

import numpy as np
import pytest

from foreign import fake_tsim, to_foreign
from oracle import oracle_np as O
from tsim_amd import prng, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", [("C2", dict()), ("C2", dict(approx=True)), ("C4", dict()), ("C5", dict(physical=False))])
def test_foreign_program_through_the_seam(hip, name, kw):
    prog, cfg = synth.config_program(name, **kw)
    foreign = to_foreign(prog)
    f = synth.synth_f(1500, cfg["num_f"], cfg["p_bit"] * 2, seed=8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = O.sample_program(prog, f[:300], (5, 6))
        got = hip.sample_program(foreign, f, (5, 6))  # frozen dataclass: the handle is cached on the object
        again = hip.sample_program(foreign, f, np.array([5, 6], np.uint32))
        plain = hip.sample_program(prog, f, (5, 6))
    assert got.dtype == np.bool_ and got.shape == (1500, prog.num_outputs)
    np.testing.assert_array_equal(got[:300], want)
    np.testing.assert_array_equal(got, plain)
    np.testing.assert_array_equal(got, again)
    assert "_backend_cache" in vars(foreign)


def test_install_on_fake_tsim_runs_the_hip_backend(hip):
    """The reference's seeded KAT (test/unit/test_sampler.py:223-233) through a call site of a fake tsim package
    after install(): 48, 53, 52, 50."""
    from tsim_amd import install

    prog = to_foreign(synth.kat_h_m())
    with fake_tsim() as smod:
        prev = install.install()
        k = prng.key(0)
        counts = []
        for _ in range(4):
            k, sub = prng.split(k)
            counts.append(int(np.asarray(smod.run(prog, np.zeros((100, 0), np.uint8), sub)).sum()))
        install.uninstall(prev)
    assert counts == [48, 53, 52, 50]


def test_foreign_level_through_the_evaluate_seam(hip):
    prog, _ = synth.config_program("C2")
    foreign = to_foreign(prog)
    lv_f, lv = foreign.components[0].compiled_scalar_graphs[3], prog.components[0].compiled_scalar_graphs[3]
    pv = (np.random.default_rng(0).random((200, lv.n_params)) < 0.3).astype(np.uint8)
    z = hip.evaluate(lv_f, pv)
    np.testing.assert_array_equal(z.view(np.float32), O.evaluate(lv, pv).view(np.float32))


def test_c1_direct_only_config_through_sample_program(hip):
    """BASELINE configs[0] shape (d = 3 surface code, Clifford only: every output direct, no component),
    1000 shots through the seam function and the kernels (sampler.py:136-145,164-166)."""
    prog, cfg = synth.config_program("C1")
    assert not prog.components and prog.num_outputs == 24
    f = synth.synth_f(cfg["shots"], cfg["num_f"], cfg["p_bit"] * 10, seed=cfg["seed"])
    out = hip.sample_program(prog, f, (0, 7))
    assert out.shape == (1000, 24) and out.dtype == np.bool_
    np.testing.assert_array_equal(out, f.astype(bool))
    np.testing.assert_array_equal(out, O.sample_program(prog, f, (0, 7)))
    # flips + shuffled columns + a non-identity f selection, still no component
    p2 = synth.synth_program(num_f=40, n_direct=24, components=[], seed=1, shuffle_outputs=True,
                             direct_flip_fraction=0.4, identity_direct=False)
    f2 = synth.synth_f(1000, 40, 0.1, seed=2)
    got = hip.sample_program(to_foreign(p2), f2, (0, 7))
    np.testing.assert_array_equal(got, O.sample_program(p2, f2, (0, 7)))
    hp = hip.get_hip_program(p2)
    packed, _ = hp.sample_batch(f2, (0, 7), bit_packed=True)
    np.testing.assert_array_equal(packed[:, :3], np.packbits(got, axis=1, bitorder="little"))


@pytest.mark.parametrize("tables", [True, False])
def test_every_leaf_type_on_the_gpu(hip, tables):
    """A foreign program with 0-d scalar leaves, no output_reindex and a ZERO-graph level (compile/evaluate.py:34-35:
    amplitude 0): the kernels give the oracle's bits and its normalisation deviation - exactly 1, which the seam
    function turns into the reference's ValueError (sampler.py:149-161)."""
    from test_foreign_program import _program_with_an_empty_level
    from oracle import oracle_c as OC

    prog = _program_with_an_empty_level()
    foreign = to_foreign(prog, scalar_leaves=True)
    from tsim_amd.program import from_tsim

    hp = hip.HipProgram(from_tsim(foreign), pattern_tables=tables)
    f = synth.synth_f(3000, 20, 0.1, seed=4)
    want, wdev = OC.OracleProgram(prog).sample_program(f, (1, 2), return_devs=True)
    got, gdev = hp.sample_batch(f, (1, 2))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32))
    assert float(gdev[0]) == 1.0
    with pytest.raises(ValueError):
        hip.sample_program(foreign, f, (1, 2))
    # the same level through the evaluate seam: zeros
    z = hip.evaluate(foreign.components[0].compiled_scalar_graphs[1], f[:50, :7])
    assert z.shape == (50,) and not z.any()
