"""Host-side cost of one pipelined launch: tiny batches (the GPU is never the bottleneck), many begin/end calls."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
prog, cfg = synth.config_program(name)
hp = backend.HipProgram(prog)
B, nf = 4096, cfg["num_f"]
wf = (nf + 63) // 64
f = synth.synth_f(B, nf, cfg["p_bit"], seed=1)
pk = np.zeros((B, wf * 8), np.uint8)
q = np.packbits(f, axis=1, bitorder="little"); pk[:, :q.shape[1]] = q
d_f = hp.malloc(pk.nbytes); hp.h2d(d_f, pk)
rb = (prog.num_outputs + 7) // 8
outs = [hp.malloc(B * rb + 16) for _ in range(16)]
def run(n):
    for i in range(n):
        s = i % 16
        hp.sample_batch_device_begin(s, d_f.ptr, B, nf, (1, i), outs[s].ptr, shot_offset=B * (i + 1), inputs_ready=True, out_bit_packed=True)
    for s in range(16):
        hp.sample_batch_device_end(s)
    hp.synchronize()
run(200)
for rep in range(3):
    t = time.perf_counter(); run(4000); dt = time.perf_counter() - t
    print(name, "us per launch (host, incl. waits):", round(dt / 4000 * 1e6, 2))
