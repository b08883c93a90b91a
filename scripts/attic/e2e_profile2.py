"""Host-side timeline of one CompiledDetectorSampler.sample(noise="device", bit_packed=True) call."""
import sys, time, cProfile, pstats
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import warnings
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
nf = cfg["num_f"]
s = CompiledDetectorSampler(prog, channel_probs=[error_probs(cfg["p_bit"])] * nf, error_transform=np.eye(nf, dtype=np.uint8), seed=1, noise="device")
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
for _ in range(3): s.sample(shots, batch_size=1_000_000, append_observables=True, bit_packed=True)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); s.sample(shots, batch_size=1_000_000, append_observables=True, bit_packed=True); ts.append(time.perf_counter() - t0)
print("sample(): median %.3f ms  -> %.3e shots/s" % (sorted(ts)[2] * 1e3, shots / sorted(ts)[2]))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): s.sample(shots, batch_size=1_000_000, append_observables=True, bit_packed=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
