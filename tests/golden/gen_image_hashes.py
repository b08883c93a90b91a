#!/usr/bin/env python3
"""The packer's image, hashed: ``tsim_program_finalize`` packs a program into the device image on the HOST before it looks for a
GPU, and ``TSIM_AMD_DEBUG=finalize,imghash`` prints an FNV-1a hash of the image after every phase.  This script records those
hashes for a set of programs (BASELINE configurations, known-answer programs, shape classes, fuzzed programs; both formulations)
in ``image_hashes.json``; ``tests/test_packer_image.py`` compares.  A change of the image LAYOUT is a reason to run this again -
a rewrite of the packer's algebra (round 6: GF(2) forms at word level, term-table power tables, chunk values by increments) is not:
the hashes must not move.

    python tests/golden/gen_image_hashes.py            # rewrite the golden file
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_hashes.json")

CHILD = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from tsim_amd import backend, synth
import test_gpu_fuzz as F
progs = []
for n in ("C2", "C3", "C4", "C5"):
    progs.append((n, synth.config_program(n)[0]))
    progs.append((n + "-approx", synth.config_program(n, approx=True)[0]))
for n in ("n9", "n16", "n24", "3narrow", "F60", "F70", "2wide", "20narrow", "w12"):
    progs.append((n, synth.shape_class_program(n)[0]))
for k in ("kat_h_m", "kat_t_gate", "kat_r_gate", "kat_bell", "kat_x_error_detector"):
    progs.append((k, getattr(synth, k)()))
for seed in range(12):
    progs.append(("fuzz%%d" %% seed, F.random_program(np.random.default_rng(seed))[0]))
    r = F.random_physical_program(np.random.default_rng(seed))
    progs.append(("fuzzp%%d" %% seed, r[0] if isinstance(r, tuple) else r))
for name, prog in progs:
    for mode in ("auto", "faithful"):
        sys.stderr.write("## %%s %%s\n" %% (name, mode)); sys.stderr.flush()
        try:
            backend.HipProgram(prog, mode=mode)
        except Exception:
            pass  # (no GPU here: finalize fails AFTER the host phases; on a GPU box it succeeds - same marks either way)
"""


def collect() -> dict:
    env = dict(os.environ, TSIM_AMD_DEBUG="finalize,imghash")
    env.pop("TSIM_AMD_TUNE", None)
    env.pop("TSIM_AMD_MODE", None)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=1200)
    res: dict = {}
    cur = None
    for line in r.stderr.splitlines():
        if line.startswith("## "):
            cur = line[3:].strip()
            res[cur] = {}
            continue
        m = re.match(r"\[tsim\] finalize: (\S.*?) [0-9.]+ ms \(image (\d+) words, hash ([0-9a-f]+)\)", line)
        if m and cur is not None and not m.group(1).startswith(" "):
            # the image after the three host phases (indented marks are sub-phases of the same image state)
            # (the third host phase, the pattern-table plan, may follow the device's free memory: not pinned)
            if m.group(1) in ("rows / fast formulation packed", "chunk / column tables"):
                res[cur].setdefault(m.group(1), [int(m.group(2)), m.group(3)])
    return res


if __name__ == "__main__":
    h = collect()
    with open(OUT, "w") as f:
        json.dump(h, f, indent=0, sort_keys=True)
    print(f"{len(h)} programs x phases -> {OUT}")
