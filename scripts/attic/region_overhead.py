"""Where the fixed ~70 us of a short timed region go (bench.py --steps 20 is 288 us for 217 us of steady-state work):
host time inside the steps call, wait in synchronize(), and an empty synchronize, for regions of K steps."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
prog, cfg = synth.config_program(name)
hp = backend.HipProgram(prog)
B, nf = 1_000_000, cfg["num_f"]
wf = (nf + 63) // 64
bufs = []
for k in range(4):
    f = synth.synth_f(B, nf, cfg["p_bit"], seed=k)
    pk = np.packbits(f, axis=1, bitorder="little")
    pk = np.ascontiguousarray(np.pad(pk, ((0, 0), (0, wf * 8 - pk.shape[1]))))
    d = hp.malloc(pk.nbytes); hp.h2d(d, pk); bufs.append(d)
outs = [hp.malloc(B * 8) for _ in range(32)]
ks = (C.c_uint32 * 2)(1, 2)
def arrs(k):
    return ((C.c_void_p * k)(*[bufs[i % 4].ptr for i in range(k)]), (C.c_void_p * k)(*[outs[i % 32].ptr for i in range(k)]))
def run(k, a):
    hp.sample_steps_device(a[0], B, nf, ks, a[1], inputs_ready=True, out_bit_packed=True)
for _ in range(8):
    run(8, arrs(8)); hp.synchronize()
a64 = arrs(64)
for K in [int(x) for x in (sys.argv[2:] or [1, 2, 4, 8, 16, 20, 24, 40])]:
    a = arrs(K)
    res = []
    for rep in range(12):
        t_end = time.perf_counter() + 0.03
        while time.perf_counter() < t_end:  # sustained clocks
            run(64, a64); hp.synchronize()
        t0 = time.perf_counter(); run(K, a); t1 = time.perf_counter(); hp.synchronize(); t2 = time.perf_counter(); hp.synchronize(); t3 = time.perf_counter()
        res.append((t1 - t0, t2 - t1, t3 - t2))
    res = np.array(res[2:]) * 1e6
    m = np.median(res, axis=0)
    print(f"K {K:3d}: total {m[0] + m[1]:7.1f} us ({(m[0] + m[1]) / K:6.2f} per step)  enqueue {m[0]:6.1f}  wait {m[1]:6.1f}  empty sync {m[2]:5.1f}")
