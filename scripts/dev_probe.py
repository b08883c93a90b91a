#!/usr/bin/env python3
"""Normalisation deviations of one shape class: row kernel (one-batch API) and the fused steps path against the C oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tsim_amd import backend, synth, prng
from oracle import oracle_c as OC
from test_gpu_steps import _run_steps, _subkeys

name = sys.argv[1]
prog, c = synth.shape_class_program(name)
nf = c["num_f"]
f = synth.synth_f(1500, nf, c["p_bit"], seed=40)
op = OC.OracleProgram(prog)
want, wdev = op.sample_program(f, (9, 10), return_devs=True)
print("oracle dev", wdev, "overflow flag", getattr(op, "overflow", None))
for env in ({}, {"TSIM_AMD_MODE": "faithful"}, {"TSIM_AMD_TUNE": "trie=0"}):
    os.environ.update(env)
    hp = backend.HipProgram(prog)
    got, dev = hp.sample_batch(f, (9, 10))
    print(env, "one-batch: samples equal", np.array_equal(got, want), "dev", dev, hp.path_counts(), "fast", hp.info()["fast"])
    devs = []
    key = prng.key(5)
    outs, _ = _run_steps(hp, prog, [f, f], key, nf, packed=True, devs=devs)
    _, subs = _subkeys(key, 2)
    w2, wd2 = op.sample_program(f, subs[0], return_devs=True)
    print("   steps: samples equal", np.array_equal(outs[0], np.packbits(w2, axis=1, bitorder="little")), "dev", devs[0][:1], "oracle", wd2, hp.path_counts())
    hp.close()
    for k in env: os.environ.pop(k)

# which rows of the steps path differ (default switches)
hp = backend.HipProgram(prog)
key = prng.key(5)
outs, _ = _run_steps(hp, prog, [f, f], key, nf, packed=True)
_, subs = _subkeys(key, 2)
w2 = np.packbits(op.sample_program(f, subs[0]), axis=1, bitorder="little")
bad = np.flatnonzero((outs[0] != w2).any(axis=1))
fsel = np.asarray(prog.components[0].f_selection)
wt = f[:, fsel].sum(axis=1)
print("mismatching rows", len(bad), "of", len(f), "first", bad[:10], "their f_sel weights", wt[bad[:10]], "weight histogram of all rows", np.bincount(wt))
if len(bad):
    r = bad[0]
    print("row", r, "got ", np.unpackbits(outs[0][r], bitorder="little")[:prog.num_outputs])
    print("row", r, "want", np.unpackbits(w2[r], bitorder="little")[:prog.num_outputs])
