"""The drop-in seam itself: backend.sample_program(program, f_params uint8[B, num_f], key) -> bool[B, n_out] with host arrays
in the reference's layout (what tsim.sampler.sample_program is replaced by)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth, prng
warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2"); nf = cfg["num_f"]
for B in (100_000, 1_000_000):
    f = synth.synth_f(B, nf, 0.02, seed=1)
    key = prng.key(5)
    backend.sample_program(prog, f, key)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); out = backend.sample_program(prog, f, key); ts.append(time.perf_counter() - t)
    t = sorted(ts)[2]
    print(f"sample_program B={B}: {t*1e3:.2f} ms -> {B/t:.3e} shots/s   ({f.nbytes/1e6:.0f} MB in, {np.asarray(out).nbytes/1e6:.0f} MB out)", flush=True)
    hp = backend.get_hip_program(prog)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); hp.sample_batch(f, key); ts.append(time.perf_counter() - t)
    print(f"  HipProgram.sample_batch alone: {sorted(ts)[2]*1e3:.2f} ms")
