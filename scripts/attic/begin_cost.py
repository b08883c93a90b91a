"""Host cost of tsim_sample_batch_device_begin in the steady state (C2, 1e6 shots per launch, inputs
resident): per-call wall time, split into the launches that also flush a deferred hard-row batch and
the others, and the resulting step rate without any Python work besides the call."""
import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import backend as hip, synth, prng
prog, cfg = synth.config_program("C2")
hp = hip.HipProgram(prog)
B, nf = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000000), cfg["num_f"]
f = synth.synth_f(B, nf, 0.02, seed=1)
fp = np.packbits(f, axis=1, bitorder="little"); fp = np.ascontiguousarray(np.pad(fp, ((0, 0), (0, 8 - fp.shape[1]))))
d_f = hp.malloc(B * 8); hp.h2d(d_f, fp)
outs = [hp.malloc(B * 8) for _ in range(16)]
lib, h = hp._lib, hp._h
for n in (8, 16):
    for it in range(3):
        hp.synchronize()
        t0 = time.perf_counter(); ts = []
        for j in range(400):
            t1 = time.perf_counter()
            lib.tsim_sample_batch_device_begin(h, j % n, d_f.ptr, B, nf, 1, j, 0, outs[j % n].ptr, None, None, 1)
            ts.append(time.perf_counter() - t1)
        t_enq = time.perf_counter() - t0
        hp.synchronize(); t_all = time.perf_counter() - t0
        ts = np.array(ts[40:]) * 1e6
        print(n, "slots: enqueue %.1f us/step, total %.1f us/step; begin() median %.1f p90 %.1f max %.1f; every 4th: %.1f others: %.1f"
              % (t_enq / 400 * 1e6, t_all / 400 * 1e6, np.median(ts), np.percentile(ts, 90), ts.max(), np.median(ts[3::4]),
                 np.median(np.concatenate([ts[0::4], ts[1::4], ts[2::4]]))))
