"""Host-side cost of one pipelined step (tiny batch, so GPU work is negligible)."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, prng, synth

prog, cfg = synth.config_program("C2")
hp = backend.HipProgram(prog)
B, nf = 1000, cfg["num_f"]
d_f = hp.malloc(B * 8); d_o = [hp.malloc(B * 8) for _ in range(4)]
hp.h2d(d_f, np.zeros((B, 8), np.uint8))
key = prng.key(1)
def t(fn, n=2000):
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    hp.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
k = [key]
def f_split(i): k[0], _ = prng.split(k[0])
print("prng.split          %.1f us" % t(f_split))
print("serial launch       %.1f us" % t(lambda i: hp.sample_batch_device(d_f.ptr, B, nf, key, d_o[0].ptr)))
def f_pipe(i):
    b = i % 4
    hp.sample_batch_device_end(b)
    hp.sample_batch_device_begin(b, d_f.ptr, B, nf, key, d_o[b].ptr)
print("begin+end           %.1f us" % t(f_pipe))
hp.profile_enable(True)
print("begin+end profiling %.1f us" % t(f_pipe))
hp.profile_read(reset=True); hp.profile_enable(False)
hp2 = backend.HipProgram(prog, pattern_tables=False)
print("serial, no tables   %.1f us" % t(lambda i: hp2.sample_batch_device(d_f.ptr, B, nf, key, d_o[0].ptr)))
