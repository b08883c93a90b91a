"""sample() end to end as a function of batch_size (C2, 4e6 shots, bit_packed and bools)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2"); nf = cfg["num_f"]
n = 4_000_000
for noise in ("device", "host"):
    s = CompiledDetectorSampler(prog, channel_probs=[error_probs(0.02)] * nf, error_transform=np.eye(nf, dtype=np.uint8), seed=1, noise=noise)
    for bs in (10_000, 100_000, 1_000_000, None):
        for kw in ({"bit_packed": True}, {}):
            s.sample(n, batch_size=bs, **kw)
            t = time.perf_counter(); s.sample(n, batch_size=bs, **kw); dt = time.perf_counter() - t
            print(f"noise={noise:6s} batch_size={str(bs):8s} {'bit_packed' if kw else 'bools     '} {dt*1e3:8.2f} ms -> {n/dt:.3e} shots/s", flush=True)
