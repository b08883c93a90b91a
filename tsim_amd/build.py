"""In-tree build of ``libtsim_hip.so`` for gfx950 (``python -m tsim_amd.build``).

``hipcc`` cross-compiles without a GPU.  The library is several translation units (``csrc/*.hip``,
``csrc/*.cpp``) compiled in parallel to objects under ``tsim_amd/_build/`` and linked to
``tsim_amd/libtsim_hip.so``, which travels with the source tree (git-ignored, not gpurun-ignored).

Staleness is decided by CONTENT, not by mtime: the SHA-256 of every source, header and flag is stored
next to the library (``libtsim_hip.so.sha256``) and per object; a snapshot of the tree whose sources
differ from what the shipped binary was built from is rebuilt where it lands (``hipcc`` is part of the
image on the GPU box too), so a stale ``.so`` can never be used silently.
"""

from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
BUILD = HERE / "_build"
OUT = HERE / "libtsim_hip.so"
STAMP = HERE / "libtsim_hip.so.sha256"

HIP_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",  # float32 epilogue must round once per operation
    "-fPIC",
]
CXX_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-msse4.1"]  # sse4.1: ceil() as one roundsd in tsim_pcg.cpp
LINK_LIBS = ["-lrccl"]  # tsim_dist.hip: the gather of the detector rows over xGMI


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cpp"))


def _headers() -> list[Path]:
    return sorted(CSRC.glob("*.h")) + [HERE.parent / "include" / "tsim_hip.h"]


def _extra_flags() -> list[str]:
    return os.environ.get("TSIM_AMD_EXTRA_FLAGS", "").split()


def _digest(paths: list[Path], flags: list[str]) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(p.name.encode())
        h.update(b"\0")
        h.update(p.read_bytes())
        h.update(b"\0")
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def source_digest() -> str:
    """Hash of everything the library is built from."""
    return _digest(_sources() + _headers(), HIP_FLAGS + CXX_FLAGS + LINK_LIBS + _extra_flags())


def needs_build() -> bool:
    if not OUT.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != source_digest()


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libtsim_hip.so")
    return hipcc


def _compile_one(src: Path, verbose: bool) -> Path:
    """Compile one translation unit unless its object is up to date (content hash of src + headers)."""
    hip = src.suffix == ".hip"
    flags = (HIP_FLAGS if hip else CXX_FLAGS) + _extra_flags()
    obj = BUILD / (src.name + ".o")
    stamp = BUILD / (src.name + ".sha256")
    want = _digest([src] + _headers(), flags)
    if obj.exists() and stamp.exists() and stamp.read_text().strip() == want:
        return obj
    cmd = [_hipcc(), *flags, "-c", str(src), "-o", str(obj)]
    if not hip:
        cmd.insert(1, "-x")
        cmd.insert(2, "c++")
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(want)
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    BUILD.mkdir(exist_ok=True)
    # one builder at a time: the ranks of a multi-process launch that all find a stale binary must not compile and
    # link into the same files side by side - the first one builds, the others wait and find it up to date
    with open(BUILD / ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return OUT
        return _build_locked(force, verbose)


def _build_locked(force: bool, verbose: bool) -> Path:
    if force:
        for f in BUILD.glob("*.sha256"):
            f.unlink()
    srcs = _sources()
    jobs = max(1, min(len(srcs), (os.cpu_count() or 2)))
    with ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(lambda s: _compile_one(s, verbose), srcs))
    tmp = OUT.with_suffix(".so.tmp")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(tmp), *LINK_LIBS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)
    STAMP.write_text(source_digest())
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
