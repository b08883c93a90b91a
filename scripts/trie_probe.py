#!/usr/bin/env python3
"""Prefix-tree pattern tables (tsim_trie.hip.h) on one shape class: which rows miss, per-batch path counts (round 6 probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tsim_amd import backend, synth, prng
from test_gpu_steps import _run_steps

name = sys.argv[1] if len(sys.argv) > 1 else "n13"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
prog, c = synth.shape_class_program(name)
hp = backend.HipProgram(prog)
print(name, hp.info())
nf = c["num_f"]
fs = [synth.synth_f(B, nf, c["p_bit"], seed=900 + i) for i in range(4)]
for rep in range(6):
    hp.path_counts(reset=True)
    t0 = time.perf_counter()
    _run_steps(hp, prog, fs, prng.key(rep), nf, packed=True)
    print(rep, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), hp.path_counts(), "pending", hp.info()["pattern_build_pending"], hp.info()["pattern_max_weight"])
    time.sleep(0.05)
