"""Exporter path on a duck-typed foreign program (CPU part): from_tsim / save_npz / install()."""

import numpy as np
import pytest

from foreign import ForeignArray, fake_tsim, to_foreign
from oracle import oracle_np as O
from tsim_amd import synth
from tsim_amd.program import from_tsim, load_npz, save_npz, validate_program


def _same_program(a, b):
    assert a.num_outputs == b.num_outputs and a.num_detectors == b.num_detectors
    np.testing.assert_array_equal(a.direct_f_indices, b.direct_f_indices)
    np.testing.assert_array_equal(a.direct_flips, b.direct_flips)
    np.testing.assert_array_equal(a.output_order, b.output_order)
    assert (a.output_reindex is None) == (b.output_reindex is None)
    for ca, cb in zip(a.components, b.components):
        assert tuple(ca.output_indices) == tuple(cb.output_indices)
        np.testing.assert_array_equal(ca.f_selection, cb.f_selection)
        for la, lb in zip(ca.compiled_scalar_graphs, cb.compiled_scalar_graphs):
            for fam in ("node_phases", "halfpi_phases", "pi_products", "phase_pairs", "prefactor"):
                for k, v in vars(getattr(la, fam)).items():
                    w = getattr(getattr(lb, fam), k)
                    if isinstance(v, np.ndarray):
                        assert isinstance(w, np.ndarray) and w.dtype == v.dtype and w.flags.c_contiguous
                        np.testing.assert_array_equal(v, w)
                    else:
                        assert v == w


@pytest.mark.parametrize("kw", [dict(), dict(approx=True), dict(physical=False)])
def test_from_tsim_converts_foreign_containers(kw, tmp_path):
    prog, cfg = synth.config_program("C2", **kw)
    foreign = to_foreign(prog)
    assert not isinstance(foreign.direct_f_indices, np.ndarray)
    back = from_tsim(foreign)
    validate_program(back, cfg["num_f"])
    _same_program(prog, back)
    # and through the .npz exporter, as it would run where tsim is installed
    save_npz(tmp_path / "p.npz", foreign, error_transform=np.eye(3, dtype=np.uint8))
    again, extra = load_npz(tmp_path / "p.npz")
    _same_program(prog, again)
    f = synth.synth_f(64, cfg["num_f"], 0.05, seed=2)
    np.testing.assert_array_equal(O.sample_program(prog, f, (1, 2)), O.sample_program(again, f, (1, 2)))


def _program_with_an_empty_level():
    """A component whose second level has ZERO graphs (compile/evaluate.py:34-35: amplitude 0 for every row): outputs
    in order (output_reindex None), scalar leaves, approximate flag on one level only."""
    from tsim_amd.program import CompiledComponent, empty_scalar_graphs, make_program

    base = synth.physical_program(num_f=20, n_direct=3, components=[dict(n=2, F=6, G=[2, 3, 4])], seed=11)
    comp = base.components[0]
    levels = list(comp.compiled_scalar_graphs)
    levels[1] = empty_scalar_graphs(levels[1].n_params)
    comp2 = CompiledComponent(comp.output_indices, comp.f_selection, tuple(levels))
    direct = [(int(base.output_order[j]), int(base.direct_f_indices[j]), bool(base.direct_flips[j])) for j in range(3)]
    return make_program([comp2], direct, base.num_outputs, base.num_detectors)


def test_every_leaf_type_the_reference_produces(tmp_path):
    """Static ints / bools as 0-d array-likes, non-numpy array leaves, a None output_reindex, a zero-graph level:
    from_tsim, the .npz exporter of this package and the dependency-free one of scripts/export_from_tsim.py agree."""
    import importlib.util
    import os

    prog = _program_with_an_empty_level()
    assert prog.output_reindex is None and prog.components[0].compiled_scalar_graphs[1].num_graphs == 0
    foreign = to_foreign(prog, scalar_leaves=True)
    back = from_tsim(foreign)
    validate_program(back, 20)
    _same_program(prog, back)
    assert type(back.num_outputs) is int and type(back.components[0].compiled_scalar_graphs[0].num_graphs) is int
    assert type(back.components[0].compiled_scalar_graphs[0].prefactor.has_approximate_floatfactors) is bool
    spec = importlib.util.spec_from_file_location(
        "export_from_tsim", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "export_from_tsim.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.save_npz_plain(tmp_path / "plain.npz", foreign, n_channels=np.int64(0), error_transform=np.zeros((20, 0), np.uint8))
    save_npz(tmp_path / "ours.npz", foreign, n_channels=np.int64(0), error_transform=np.zeros((20, 0), np.uint8))
    a, xa = load_npz(tmp_path / "plain.npz")
    b, xb = load_npz(tmp_path / "ours.npz")
    _same_program(prog, a)
    _same_program(prog, b)
    assert sorted(xa) == sorted(xb)
    # the oracle on the empty level: amplitude 0 -> the output is never 1 and the normalisation check sees a vanishing
    # marginal (deviation 1, what the reference turns into a ValueError, sampler.py:149-161)
    f = synth.synth_f(200, 20, 0.1, seed=4)
    bits, devs = O.sample_program(prog, f, (1, 2), return_devs=True)
    col = prog.components[0].output_indices[0]
    assert not bits[:, col].any() and float(devs[0]) == 1.0


def test_from_tsim_shuffled_outputs_and_reindex():
    prog = synth.physical_program(num_f=30, n_direct=5, components=[dict(n=2, F=8, G=[2, 3, 4])], seed=3,
                                  shuffle_outputs=True, identity_direct=False, direct_flip_fraction=0.5)
    assert prog.output_reindex is not None
    _same_program(prog, from_tsim(to_foreign(prog)))


def test_install_rebinds_the_module_global():
    """install() on a fake tsim package: call sites that resolve tsim.sampler.sample_program at call time reach
    the backend (which fails loudly here: no GPU), uninstall() restores the previous function."""
    from tsim_amd import _lib, backend, install

    prog = to_foreign(synth.kat_h_m())
    with fake_tsim() as smod:
        with pytest.raises(RuntimeError, match="not installed"):
            smod.run(prog, np.zeros((4, 0), np.uint8), (0, 1))
        prev = install.install(patch_evaluate=True)
        assert smod.sample_program is not prev and smod.evaluate is backend.evaluate
        if _lib.device_count() == 0:
            with pytest.raises(_lib.HipBackendError):
                smod.run(prog, np.zeros((4, 0), np.uint8), (0, 1))
        install.uninstall(prev)
        assert smod.sample_program is prev
    with pytest.raises(RuntimeError, match="tsim is not importable"):
        install.install()
