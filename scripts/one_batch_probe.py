import time, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tsim_amd import backend as hip, synth, prng
for name in sys.argv[1:]:
    prog, c = synth.shape_class_program(name)
    nf = c["num_f"]
    B = 1_000_000
    hp = hip.HipProgram(prog)
    f = synth.synth_f(B, nf, c["p_bit"], seed=1)
    fp = np.packbits(f, axis=1, bitorder="little")
    wf = max(1, (nf + 63) // 64)
    fw = np.zeros((B, wf * 8), np.uint8); fw[:, : fp.shape[1]] = fp
    d_f = hp.malloc(fw.nbytes); hp.h2d(d_f, fw)
    wo = (hp.num_outputs + 63) // 64
    d_o = hp.malloc(B * wo * 8)
    for rep in range(3):
        t0 = time.perf_counter()
        n = 10
        for i in range(n):
            hp.sample_batch_device(d_f.ptr, B, nf, prng.key(i), d_o.ptr)
        hp.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(name, os.environ.get("TSIM_AMD_TUNE", ""), "one-batch API: %.1f us per 1e6 shots" % (dt * 1e6), hp.path_counts(reset=True), flush=True)
    hp.close()
