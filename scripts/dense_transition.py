"""Sparse -> dense -> sparse batches on ONE handle through tsim_sample_steps_device: ms per step of every call (the
launch plan follows the hard-row counts of earlier launches; bench.py's `dense` leg does the same)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth
prog, cfg = synth.config_program("C2")
hp = backend.HipProgram(prog)
B, nf = 1_000_000, cfg["num_f"]
def dev_f(p, seed):
    f = synth.synth_f(B, nf, p, seed=seed)
    pk = np.ascontiguousarray(np.packbits(f, axis=1, bitorder="little")).view(np.uint64)
    d = hp.malloc(pk.nbytes); hp.h2d(d, pk); return d
sparse = [dev_f(0.02, k) for k in range(2)]
dense = [dev_f(float(sys.argv[1]) if len(sys.argv) > 1 else 0.1, 10 + k) for k in range(2)]
outs = [hp.malloc(B * 8) for _ in range(16)]
ks = (C.c_uint32 * 2)(1, 2)
j = 0
def run(fl, n, label):
    global j
    t0 = time.perf_counter()
    hp.sample_steps_device([fl[i % 2].ptr for i in range(n)], B, nf, ks, [outs[(j + i) % 16].ptr for i in range(n)], inputs_ready=True, out_bit_packed=True)
    hp.synchronize()
    j += n
    print(f"{label}: {n} steps, {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step", flush=True)
for r in range(4): run(sparse, 4, "sparse")
for r in range(8): run(dense, 4, "dense ")
run(dense, 40, "dense ")
for r in range(4): run(sparse, 4, "sparse")
run(sparse, 40, "sparse")
