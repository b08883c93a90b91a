"""In-tree build of ``libtsim_hip.so`` for gfx950 (``python -m tsim_amd.build``).

``hipcc`` cross-compiles without a GPU; the shared object lands next to this
package so that it travels with the source tree (it is git-ignored).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "csrc" / "tsim_hip.hip"
DEPS = sorted((HERE / "csrc").glob("*")) + [HERE.parent / "include" / "tsim_hip.h"]
OUT = HERE / "libtsim_hip.so"

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",  # float32 epilogue must round once per operation
    "-fPIC",
    "-shared",
]


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    return any(d.stat().st_mtime > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libtsim_hip.so")
    tmp = OUT.with_suffix(".so.tmp")
    extra = os.environ.get("TSIM_AMD_EXTRA_FLAGS", "").split()
    cmd = [hipcc, *FLAGS, *extra, str(SRC), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
