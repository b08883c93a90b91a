"""End-to-end sample(postselection_mask=...) on the C2 shape: shots in which a masked direct detector fires are discarded
before sample_program (sampler.py:422-545)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
prog, cfg = synth.config_program("C2")
nf = cfg["num_f"]
for noise in ("device", "host"):
    s = CompiledDetectorSampler(prog, channel_probs=[error_probs(p)] * nf, error_transform=np.eye(nf, dtype=np.uint8), seed=1, noise=noise)
    nd = s._num_detectors
    mask = np.ones(nd, dtype=bool)
    for name, kw in (("plain", {}), ("post-selected (all direct detectors)", {"postselection_mask": mask})):
        s.sample(shots, batch_size=1_000_000, bit_packed=True, **kw)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = s.sample(shots, batch_size=1_000_000, bit_packed=True, **kw); ts.append(time.perf_counter() - t)
        t = sorted(ts)[1]
        print(f"noise={noise:6s} {name:40s} {t*1e3:8.2f} ms -> {shots/t:.3e} shots/s", flush=True)
