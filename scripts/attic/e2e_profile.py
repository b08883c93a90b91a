"""cProfile of one steady-state CompiledDetectorSampler.sample() call per mode (where does the host time go?)."""
import cProfile, pstats, sys, time, warnings
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
probs = [error_probs(cfg["p_bit"])] * cfg["num_f"]
T = np.eye(cfg["num_f"], dtype=np.uint8)
shots, batch = 4_000_000, 1_000_000
for noise, kw in (("device", dict(append_observables=True)), ("host", dict(append_observables=True)),
                  ("device", dict(append_observables=True, bit_packed=True))):
    s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=1, noise=noise)
    for _ in range(2):
        s.sample(shots, batch_size=batch, **kw)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); s.sample(shots, batch_size=batch, **kw); pr.disable()
    print(f"=== noise={noise} {kw}: {(time.perf_counter()-t0)*1e3:.1f} ms")
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
