"""Which pipeline slots a step sequence rotates through vs throughput (C2, 1e6 shots per launch)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tsim_amd import backend, synth
prog, cfg = synth.config_program("C2")
hp = backend.HipProgram(prog)
B, nf = 1_000_000, cfg["num_f"]
fs = []
for k in range(4):
    f = synth.synth_f(B, nf, cfg["p_bit"], seed=1 + k)
    pk = np.zeros((B, 8), np.uint8); q = np.packbits(f, axis=1, bitorder="little"); pk[:, :q.shape[1]] = q
    d = hp.malloc(pk.nbytes); hp.h2d(d, pk); fs.append(d)
outs = [hp.malloc(B * 8) for _ in range(16)]
def run(order, n):
    for i in range(n):
        s = order[i % len(order)]
        hp.sample_batch_device_begin(s, fs[i % 4].ptr, B, nf, (1, i), outs[s].ptr, inputs_ready=True, out_bit_packed=True)
    for s in range(16):
        hp.sample_batch_device_end(s)
    hp.synchronize()
orders = {
    "0..15": list(range(16)), "0..13": list(range(14)), "2..15": list(range(2, 16)), "0..11": list(range(12)), "4..15": list(range(4, 16)),
    "0..9": list(range(10)), "6..15": list(range(6, 16)), "4..13": list(range(4, 14)),
    "16 slots, pairs swapped (1,0,3,2,..)": [i ^ 1 for i in range(16)],
    "16 slots, batch halves swapped (2,3,0,1,6,7,4,5,..)": [i ^ 2 for i in range(16)],
}
run(list(range(16)), 64)
for name, order in orders.items():
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); run(order, 400); ts.append(time.perf_counter() - t0)
    print(f"{name:55s} {1e6 * min(ts) / 400:6.2f} us/step (min of 4 x 400 steps)")
