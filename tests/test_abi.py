"""The C-ABI library loads on a GPU-less box and exports every symbol include/tsim_hip.h declares."""

import re
from pathlib import Path

import numpy as np
import pytest

from tsim_amd import _lib, synth
from tsim_amd.program import load_npz, save_npz, validate_program

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "tsim_hip.h").read_text()
    declared = set(re.findall(r"\b(tsim_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libtsim_hip.so does not export {name}"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert b"gfx950" in lib.tsim_version()


def test_argument_validation_without_gpu():
    """Errors that need no device: bad descriptions are rejected with a message (ValueError)."""
    import ctypes as C

    lib = _lib.load()
    h = C.c_void_p()
    order = np.array([0, 0], np.int32)  # not a permutation
    rc = lib.tsim_program_create(2, 0, 0, None, None, order.ctypes.data_as(C.c_void_p), C.byref(h))
    assert rc == -22 and b"permutation" in lib.tsim_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "tsim_program_create")
    order = np.array([0], np.int32)
    assert lib.tsim_program_create(1, 0, 0, None, None, order.ctypes.data_as(C.c_void_p), C.byref(h)) == 0
    oi = np.array([0], np.int32)
    assert lib.tsim_program_add_component(h, 1, oi.ctypes.data_as(C.c_void_p), 0, None, 5) == -22
    assert lib.tsim_sample_batch(h, None, 1, 0, 0, 0, 0, None, 0, None) == -1  # not finalized
    lib.tsim_program_destroy(h)


def test_product_path_fails_loudly_without_device():
    from tsim_amd import backend

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.HipBackendError):
        backend.HipProgram(synth.kat_h_m())
    with pytest.raises(_lib.HipBackendError):
        backend.sample_program(synth.kat_h_m(), np.zeros((4, 0), np.uint8), (0, 1))


def test_npz_roundtrip(tmp_path):
    prog, cfg = synth.config_program("C2", approx=True)
    validate_program(prog, cfg["num_f"])
    p = tmp_path / "c2.npz"
    save_npz(p, prog, note=np.arange(3))
    back, extra = load_npz(p)
    validate_program(back, cfg["num_f"])
    assert extra["note"].tolist() == [0, 1, 2]
    assert back.num_outputs == prog.num_outputs and len(back.components) == 1
    a, b = prog.components[0].compiled_scalar_graphs[3], back.components[0].compiled_scalar_graphs[3]
    assert np.array_equal(a.pi_products.phi_params, b.pi_products.phi_params)
    assert np.array_equal(a.prefactor.approximate_floatfactors, b.prefactor.approximate_floatfactors)
    assert b.prefactor.has_approximate_floatfactors
    from oracle import oracle_np as O

    f = synth.synth_f(50, cfg["num_f"], 0.05, seed=1)
    assert np.array_equal(O.sample_program(prog, f, (1, 2)), O.sample_program(back, f, (1, 2)))


def test_key_split_matches_python_prng():
    """tsim_key_split (host helper, no device) == tsim_amd.prng.split."""
    import ctypes as C

    from tsim_amd import _lib, prng

    lib = _lib.load()
    out = (C.c_uint32 * 4)()
    k = prng.key(0)
    for _ in range(5):
        lib.tsim_key_split(k[0], k[1], out)
        new, sub = prng.split(k)
        assert (out[0], out[1]) == new and (out[2], out[3]) == sub
        k = new
    lib.tsim_key_split(0xFFFFFFFF, 0x12345678, out)
    new, sub = prng.split((0xFFFFFFFF, 0x12345678))
    assert (out[0], out[1]) == new and (out[2], out[3]) == sub


def test_parameter_limit_is_reported():
    """More than TSIM_MAX_PARAMS = 2048 parameters per level: ENOTSUP with a message, before any device call
    (the reference's own regression sizes, test_linalg.py:115-126, end at P = 1024; up to 2048 is supported)."""
    from tsim_amd import backend
    from tsim_amd.program import CompiledComponent, empty_scalar_graphs, make_program, scalar_graphs_from_terms

    P = 2049
    lv = scalar_graphs_from_terms(P, [dict(B=[(4, list(range(P)))])])
    comp = CompiledComponent(tuple(range(P)), np.zeros(0, np.int32), (empty_scalar_graphs(0), lv))
    with pytest.raises(_lib.HipBackendError, match="TSIM_MAX_PARAMS=2048"):
        backend.HipProgram(make_program([comp], [], P, 0))
