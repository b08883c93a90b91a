"""Offline soak of the HIP path against the C oracle (test infrastructure: run by hand on the GPU box,
`python tests/fuzz_many.py N [first seed]`; not collected by pytest)."""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, warnings
warnings.simplefilter("ignore")
from tsim_amd import backend as hip, synth
from oracle import oracle_c as OC
from test_gpu_fuzz import random_physical_program, random_program, random_switches
import os
bad = skipped = 0
used = {}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first + n):
    rng = np.random.default_rng(5000 + seed)
    prog, num_f = (random_physical_program if seed % 2 else random_program)(rng)
    B = int(rng.choice([1, 64, 65, 257, 1000]))
    f = synth.synth_f(B, num_f, float(rng.choice([0.0, 0.02, 0.3])), seed=seed)
    key = (int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)))
    want, wdev, ov = OC.OracleProgram(prog).sample_program(f, key, return_devs=True, return_overflow=True)
    if ov:
        skipped += 1; continue
    for mode, pt in (("auto", None), ("auto", False), ("rows", True), ("faithful", True), ("auto", 1), ("auto", 4)):
        env = random_switches(rng)  # a random assignment of the launch-plan switches per handle: results must not depend on them
        os.environ.update(env)
        try:
            hp = hip.HipProgram(prog, mode=mode, pattern_tables=pt)
        finally:
            for k_ in env: os.environ.pop(k_, None)
        got, gdev = hp.sample_batch(f, key)
        kinds = hp.info(); used[(kinds["chunk_table_kernel"], kinds["wide_sparse_kernel"], kinds["pattern_tables"])] = used.get((kinds["chunk_table_kernel"], kinds["wide_sparse_kernel"], kinds["pattern_tables"]), 0) + 1
        if pt is None and seed % 3 == 0:  # a second launch on the same handle: the adaptive plan must not change bits
            got2, _ = hp.sample_batch(f, key)
            if not np.array_equal(got2, got):
                bad += 1; print("MISMATCH (second launch) seed", seed, mode, hp.info())
        if not (np.array_equal(got, want) and np.array_equal(np.asarray(gdev, np.float32), np.asarray(wdev, np.float32), equal_nan=True)):
            bad += 1; print("MISMATCH seed", seed, mode, pt, env, hp.info())
        hp.close()  # depth-4 tables of a wide component are gigabytes: do not wait for the collector
    if seed % 50 == 49:
        import gc; gc.collect()
print("done", n, "programs; mismatches", bad, "skipped(overflow)", skipped, "handles by (chunk, wide, tables):", used)
