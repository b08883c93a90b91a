"""End-to-end CompiledDetectorSampler.sample(noise="host"): the reference's numpy/PCG64 channel stream bit for bit."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import warnings
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
nf = cfg["num_f"]
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
for packed in (False, True):
    s = CompiledDetectorSampler(prog, channel_probs=[error_probs(cfg["p_bit"])] * nf, error_transform=np.eye(nf, dtype=np.uint8), seed=1, noise="host")
    for _ in range(2): s.sample(shots, batch_size=1_000_000, append_observables=True, bit_packed=packed)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = s.sample(shots, batch_size=1_000_000, append_observables=True, bit_packed=packed); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); s._channel_sampler.sample_packed(1_000_000, out=np.empty((1_000_000, 1), np.uint64)); tc = time.perf_counter() - t0
    print(f"host noise, bit_packed={packed}: {sorted(ts)[1]*1e3:.2f} ms -> {shots/sorted(ts)[1]:.3e} shots/s   (channel sampler alone: {tc*1e3:.2f} ms per 1e6 shots -> floor {1e6/tc:.3e})")
