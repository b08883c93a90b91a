"""ctypes binding of ``libtsim_hip.so`` (C ABI: ``include/tsim_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` /
``python -m tsim_amd.build``.  There is deliberately **no fallback**: if the
shared object is missing or cannot be loaded, importing the backend fails
loudly with ``HipBackendError`` - the product path never routes through a CPU
implementation.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libtsim_hip.so"


class HipBackendError(RuntimeError):
    """The HIP extension is missing, failed to load, or a call into it failed."""


class LevelDesc(C.Structure):
    """``tsim_level_desc`` (include/tsim_hip.h)."""

    _fields_ = [
        ("num_graphs", C.c_int32),
        ("n_params", C.c_int32),
        ("ta", C.c_int32),
        ("tb", C.c_int32),
        ("tc", C.c_int32),
        ("td", C.c_int32),
        ("a_phases", C.c_void_p),
        ("a_params", C.c_void_p),
        ("a_counts", C.c_void_p),
        ("b_coeffs", C.c_void_p),
        ("b_params", C.c_void_p),
        ("c_psi_const", C.c_void_p),
        ("c_psi_params", C.c_void_p),
        ("c_phi_const", C.c_void_p),
        ("c_phi_params", C.c_void_p),
        ("d_alpha", C.c_void_p),
        ("d_alpha_params", C.c_void_p),
        ("d_beta", C.c_void_p),
        ("d_beta_params", C.c_void_p),
        ("d_counts", C.c_void_p),
        ("phase_indices", C.c_void_p),
        ("floatfactor", C.c_void_p),
        ("power2", C.c_void_p),
        ("approx", C.c_void_p),
        ("has_approx", C.c_int32),
    ]


# every symbol include/tsim_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I32, _I64, _U32 = C.c_int32, C.c_int64, C.c_uint32
SYMBOLS: dict[str, tuple] = {
    "tsim_program_create": (C.c_int, [_I32, _I32, _I32, _P, _P, _P, C.POINTER(_P)]),
    "tsim_program_add_component": (C.c_int, [_P, _I32, _P, _I32, _P, _I32]),
    "tsim_program_add_level": (C.c_int, [_P, _I32, C.POINTER(LevelDesc)]),
    "tsim_program_set_mode": (C.c_int, [_P, _I32]),
    "tsim_program_get_mode": (C.c_int, [_P, C.POINTER(_I32)]),
    "tsim_program_set_pattern_tables": (C.c_int, [_P, _I32, _I32]),
    "tsim_program_pattern_table_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I64), C.POINTER(_I32)]),
    "tsim_program_tables_pending": (C.c_int, [_P, C.POINTER(_I32)]),
    "tsim_program_finalize": (C.c_int, [_P, _I32]),
    "tsim_program_destroy": (None, [_P]),
    "tsim_sample_batch": (C.c_int, [_P, _P, _I64, _I32, _U32, _U32, _I64, _P, _I32, _P]),
    "tsim_sample_batch_device": (C.c_int, [_P, _P, _I64, _I32, _U32, _U32, _I64, _P, _P, _P]),
    "tsim_sample_batch_device_begin": (C.c_int, [_P, _I32, _P, _I64, _I32, _U32, _U32, _I64, _P, _P, _P, _U32]),
    "tsim_sample_batch_device_end": (C.c_int, [_P, _I32, _P]),
    "tsim_pipeline_wait_slot": (C.c_int, [_P, _I32, _P]),
    "tsim_pipeline_wait_stream": (C.c_int, [_P, _P]),
    "tsim_pipeline_join": (C.c_int, [_P, _P]),
    "tsim_pipeline_lane_stream": (C.c_int, [_P, _I32, C.POINTER(C.c_void_p)]),
    "tsim_sample_batch_device_compact": (C.c_int, [_P, _I32, _P, _I64, _I32, _P, _P]),
    "tsim_pipeline_set_compact_output": (C.c_int, [_P, _I32, _P]),
    "tsim_pipeline_set_compact_series": (C.c_int, [_P, _P, _I64, _I32]),
    "tsim_postselect_device": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "tsim_sample_rows_device": (C.c_int, [_P, _P, _I64, _I32, _U32, _U32, _I64, _P, _P, _P, _P, _P]),
    "tsim_evaluate": (C.c_int, [_P, _I32, _I32, _P, _I64, _P, _P, _P, _P]),
    "tsim_pack_bits_device": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "tsim_unpack_bits_device": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "tsim_compact_rows_device": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "tsim_postselect_rows_device": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P]),
    "tsim_arrange_rows_device": (C.c_int, [_P, _P, _I64, _I32, _P, _I32, _I32, _P, _P]),
    "tsim_survivors_append_device": (C.c_int, [_P, _P, _I64, C.c_uint32, _P, _P, _P, _P]),
    "tsim_gather_rows_device": (C.c_int, [_P, _P, _I32, _P, _I64, _I64, _P, _P]),
    "tsim_scatter_rows_device": (C.c_int, [_P, _P, _I32, _P, _I64, _P, _P]),
    "tsim_mem_info": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "tsim_pcg_draw": (C.c_int, [_P, _I32, C.c_double, _I64, _P]),
    "tsim_pcg_sample_channels": (C.c_int, [_P, _I32, _P, _P, _P, _P, _I32, _I64, _P, _I32]),
    "tsim_noise_create": (C.c_int, [_P, _I32, _I32, _P, _P, _P, _P, C.POINTER(_P)]),
    "tsim_noise_sample_device": (C.c_int, [_P, _I64, _U32, _U32, _P, _P]),
    "tsim_noise_destroy": (None, [_P]),
    "tsim_dist_unique_id": (C.c_int, [_P]),
    "tsim_dist_init": (C.c_int, [_I32, _P, _I32, _I32, C.POINTER(_P)]),
    "tsim_dist_destroy": (None, [_P]),
    "tsim_dist_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "tsim_dist_gather_rows": (C.c_int, [_P, _P, _I64, _P, _I32, _P]),
    "tsim_dist_alltoall_rows": (C.c_int, [_P, _P, _P, _I64, _P]),
    "tsim_dist_allreduce_max": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "tsim_dist_barrier": (C.c_int, [_P]),
    "tsim_dist_stream_wait": (C.c_int, [_P, _P, _P]),
    "tsim_dist_mark": (C.c_int, [_P, _I32, _P]),
    "tsim_dist_wait_mark": (C.c_int, [_P, _I32, _P]),
    "tsim_device_synchronize": (C.c_int, [_I32]),
    "tsim_device_count": (C.c_int, [C.POINTER(_I32)]),
    "tsim_malloc_device": (C.c_int, [_P, _I64, C.POINTER(_P)]),
    "tsim_free_device": (C.c_int, [_P, _P]),
    "tsim_malloc_pinned": (C.c_int, [_I64, C.POINTER(_P)]),
    "tsim_free_pinned": (C.c_int, [_P]),
    "tsim_memcpy_h2d": (C.c_int, [_P, _P, _P, _I64]),
    "tsim_memcpy_d2h": (C.c_int, [_P, _P, _P, _I64]),
    "tsim_synchronize": (C.c_int, [_P]),
    "tsim_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "tsim_profile_enable": (C.c_int, [_P, _I32]),
    "tsim_profile_set_sampling": (C.c_int, [_P, _I32]),
    "tsim_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
    "tsim_profile_read_stages": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "tsim_program_info": (
        C.c_int,
        [_P, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)],
    ),
    "tsim_program_stats": (C.c_int, [_P, C.POINTER(_I64)]),
    "tsim_program_path_counts": (C.c_int, [_P, C.POINTER(_I64), C.c_int32]),
    "tsim_key_split": (None, [_U32, _U32, C.POINTER(_U32)]),
    "tsim_sample_batch_device_begin_split": (C.c_int, [_P, _I32, _P, _I64, _I32, C.POINTER(_U32), _I64, _P, _P, _P, _U32]),
    "tsim_sample_steps_device": (C.c_int, [_P, _I32, C.POINTER(_P), _I64, _I32, C.POINTER(_U32), _I64, C.POINTER(_P), C.POINTER(_P), _U32]),
    "tsim_sample_steps_noise_device": (C.c_int, [_P, _P, _I32, C.POINTER(_P), _I64, _I32, C.POINTER(_U32), C.POINTER(_U32), _I64, C.POINTER(_P),
                                                 C.POINTER(_P), _U32]),
    "tsim_profile_read_steps": (C.c_int, [_P, C.POINTER(_I64), _I32]),
    "tsim_memcpy_d2h_async": (C.c_int, [_P, _P, _P, _I64, _P]),
    "tsim_memcpy_h2d_async": (C.c_int, [_P, _P, _P, _I64, _P]),
    "tsim_stream_synchronize": (C.c_int, [_P, _P]),
    "tsim_aux_stream": (C.c_int, [_P, _I32, C.POINTER(_P)]),
    "tsim_pipeline_next_slot": (C.c_int, [_P, C.POINTER(_I32)]),
    "tsim_last_error": (C.c_char_p, []),
    "tsim_version": (C.c_char_p, []),
    "tsim_tune_keys": (C.c_char_p, []),
}

_lib = None


def load(build: bool = True) -> C.CDLL:
    """Load ``libtsim_hip.so`` and bind every declared symbol (raises if absent).

    The binary must be the one built from THIS tree: the content hash recorded at build time is compared with the
    sources (a prebuilt .so travels with snapshots of the tree).  A stale or unstamped one is rebuilt here when
    ``build`` is true (hipcc is part of the image on the GPU boxes) - never used silently.  ``build=False`` (the
    host-only channel sampler's probe) never starts a compile: it raises instead, and the caller falls back.
    ``TSIM_AMD_ALLOW_STALE=1`` opts into using a binary whose stamp is missing or different, with a warning - for
    machines without hipcc that received a library built elsewhere from the same sources."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("TSIM_AMD_LIB", LIB_PATH))
    if path == LIB_PATH:
        from . import build as _build

        if _build.needs_build():
            if path.exists() and os.environ.get("TSIM_AMD_ALLOW_STALE") == "1":
                import warnings

                warnings.warn(f"tsim_amd: using {path} although its build stamp is missing or does not match the sources "
                              "(TSIM_AMD_ALLOW_STALE=1)", stacklevel=2)
            elif not build:
                raise HipBackendError(f"{path} is missing or was built from different sources (not building from this call)")
            else:
                try:
                    _build.build(verbose=False)
                except Exception as exc:
                    raise HipBackendError(
                        f"{path} is missing or was built from different sources and cannot be rebuilt here "
                        f"({exc}); build it with `python -m tsim_amd.build` (hipcc --offload-arch=gfx950), or set "
                        "TSIM_AMD_ALLOW_STALE=1 to use it as it is - there is no CPU fallback"
                    ) from exc
    if not path.exists():
        raise HipBackendError(
            f"{path} not found - build it with `python -m tsim_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback"
        )
    try:
        lib = C.CDLL(str(path))
    except OSError as exc:  # pragma: no cover - depends on the host
        raise HipBackendError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise HipBackendError(f"{path} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().tsim_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> int:
    """Raise on a negative return code: -22 -> ValueError, others -> HipBackendError."""
    if rc >= 0:
        return rc
    msg = f"{what}: {last_error()} (code {rc})"
    if rc == -22:
        raise ValueError(msg)
    raise HipBackendError(msg)


def ptr(a: np.ndarray | None):
    """Raw data pointer of a C-contiguous array (``None`` -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = C.c_int32(0)
    rc = load().tsim_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0
