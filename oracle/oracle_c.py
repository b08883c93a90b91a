"""ctypes loader for the C oracle (``oracle/oracle.c``) - TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this.  Build with ``make -C oracle`` (done by ``__graft_entry__.build()``).
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"


class LevelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_graphs", "n_params", "ta", "tb", "tc", "td")] + [
        (n, C.c_void_p)
        for n in (
            "a_phases", "a_params", "a_counts", "b_coeffs", "b_params",
            "c_psi_const", "c_psi_params", "c_phi_const", "c_phi_params",
            "d_alpha", "d_alpha_params", "d_beta", "d_beta_params", "d_counts",
            "phase_indices", "floatfactor", "power2", "approx",
        )
    ] + [("has_approx", C.c_int32)]


_lib = None


def build(force: bool = False) -> Path:
    src = HERE / "oracle.c"
    if force or not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        lib = C.CDLL(str(LIB))
        P, I32, I64, U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
        lib.orc_program_new.restype = P
        lib.orc_program_new.argtypes = [I32, I32, P, P, P]
        lib.orc_program_add_component.restype = C.c_int
        lib.orc_program_add_component.argtypes = [P, I32, P, I32, P, I32]
        lib.orc_program_add_level.restype = C.c_int
        lib.orc_program_add_level.argtypes = [P, I32, C.POINTER(LevelDesc)]
        lib.orc_program_free.restype = None
        lib.orc_program_free.argtypes = [P]
        lib.orc_sample_program.restype = C.c_int
        lib.orc_sample_program.argtypes = [P, P, I64, I32, U32, U32, I64, P, P, I32]
        lib.orc_evaluate.restype = C.c_int
        lib.orc_evaluate.argtypes = [P, I32, I32, P, I64, P, P, P]
        lib.orc_num_threads.restype = C.c_int
        _lib = lib
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(np.asarray(a), dtype=dt)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class OracleProgram:
    """A program loaded into the C oracle."""

    def __init__(self, program):
        lib = load()
        self._lib = lib
        self.program = program
        self.num_outputs = int(program.num_outputs)
        self.n_components = len(program.components)
        dfi = _c(program.direct_f_indices, np.int32)
        flips = _c(np.asarray(program.direct_flips).astype(np.uint8), np.uint8)
        order = _c(program.output_order, np.int32)
        self._h = lib.orc_program_new(self.num_outputs, len(dfi), _ptr(dfi), _ptr(flips), _ptr(order))
        for comp in program.components:
            oi = _c(comp.output_indices, np.int32)
            fs = _c(comp.f_selection, np.int32)
            ci = lib.orc_program_add_component(self._h, len(oi), _ptr(oi), len(fs), _ptr(fs), len(comp.compiled_scalar_graphs))
            for lv in comp.compiled_scalar_graphs:
                a, b, c, d, pre = lv.node_phases, lv.halfpi_phases, lv.pi_products, lv.phase_pairs, lv.prefactor
                G = int(lv.num_graphs)
                arrs = dict(
                    a_phases=_c(a.phases, np.uint8), a_params=_c(a.params, np.uint8), a_counts=_c(a.counts, np.int32),
                    b_coeffs=_c(b.coeffs, np.uint8), b_params=_c(b.params, np.uint8),
                    c_psi_const=_c(c.psi_const, np.uint8), c_psi_params=_c(c.psi_params, np.uint8),
                    c_phi_const=_c(c.phi_const, np.uint8), c_phi_params=_c(c.phi_params, np.uint8),
                    d_alpha=_c(d.alpha, np.uint8), d_alpha_params=_c(d.alpha_params, np.uint8),
                    d_beta=_c(d.beta, np.uint8), d_beta_params=_c(d.beta_params, np.uint8), d_counts=_c(d.counts, np.int32),
                    phase_indices=_c(pre.phase_indices, np.uint8), floatfactor=_c(pre.floatfactor, np.int32),
                    power2=_c(pre.power2, np.int32), approx=_c(pre.approximate_floatfactors, np.complex64),
                )
                desc = LevelDesc()
                desc.num_graphs, desc.n_params = G, int(lv.n_params)
                desc.ta = arrs["a_phases"].shape[1] if G else 0
                desc.tb = arrs["b_coeffs"].shape[1] if G else 0
                desc.tc = arrs["c_psi_const"].shape[1] if G else 0
                desc.td = arrs["d_alpha"].shape[1] if G else 0
                for k, v in arrs.items():
                    setattr(desc, k, v.ctypes.data if v.size else None)
                desc.has_approx = 1 if pre.has_approximate_floatfactors else 0
                rc = lib.orc_program_add_level(self._h, ci, C.byref(desc))  # arrays are deep-copied
                assert rc == 0

    def __del__(self):
        try:
            if self._h:
                self._lib.orc_program_free(self._h)
                self._h = None
        except Exception:
            pass

    def sample_program(self, f_params, key, *, shot_offset: int = 0, threads: int = 0, return_devs: bool = False,
                       return_overflow: bool = False):
        f = np.ascontiguousarray((np.asarray(f_params) != 0).astype(np.uint8))
        B, num_f = f.shape
        out = np.zeros((B, self.num_outputs), np.uint8)
        devs = np.zeros(max(1, self.n_components), np.float32)
        ov = self._lib.orc_sample_program(
            self._h, _ptr(f), B, num_f, int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF,
            int(shot_offset), out.ctypes.data_as(C.c_void_p), devs.ctypes.data_as(C.c_void_p), int(threads),
        )
        res = [out.view(np.bool_)]
        if return_devs:
            res.append(devs[: self.n_components])
        if return_overflow:
            res.append(bool(ov))
        return res[0] if len(res) == 1 else tuple(res)

    def evaluate(self, component: int, level: int, param_vals, *, exact: bool = False):
        pv = np.ascontiguousarray((np.asarray(param_vals) != 0).astype(np.uint8))
        B = pv.shape[0]
        re, im = np.zeros(B, np.float32), np.zeros(B, np.float32)
        ex = np.zeros((B, 5), np.int32)
        ov = self._lib.orc_evaluate(self._h, component, level, _ptr(pv), B, re.ctypes.data_as(C.c_void_p),
                                    im.ctypes.data_as(C.c_void_p), ex.ctypes.data_as(C.c_void_p))
        z = np.empty(B, np.complex64)
        z.real, z.imag = re, im
        return (z, ex, bool(ov)) if exact else z


def num_threads() -> int:
    return int(load().orc_num_threads())
