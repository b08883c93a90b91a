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
