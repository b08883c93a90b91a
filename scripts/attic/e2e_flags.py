"""sample() keyword surface, end to end: shots/s for every combination class (C2 shape, 4e6 shots)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler, CompiledMeasurementSampler
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
prog, cfg = synth.config_program("C2"); nf = cfg["num_f"]
kwn = dict(channel_probs=[error_probs(0.02)] * nf, error_transform=np.eye(nf, dtype=np.uint8))
cases = [("default (bool detectors)", {}), ("bit_packed", dict(bit_packed=True)), ("append_observables", dict(append_observables=True)),
         ("append + bit_packed", dict(append_observables=True, bit_packed=True)), ("separate_observables", dict(separate_observables=True)),
         ("separate + bit_packed", dict(separate_observables=True, bit_packed=True)), ("prepend_observables", dict(prepend_observables=True)),
         ("detector reference", dict(use_detector_reference_sample=True)), ("det + obs reference, append, packed", dict(use_detector_reference_sample=True, use_observable_reference_sample=True, append_observables=True, bit_packed=True))]
for noise in ("device", "host"):
    s = CompiledDetectorSampler(prog, seed=1, noise=noise, **kwn)
    for name, kw in cases:
        s.sample(n // 4, batch_size=1_000_000, **kw)
        t = time.perf_counter(); s.sample(n, batch_size=1_000_000, **kw); dt = time.perf_counter() - t
        print(f"noise={noise:6s} {name:40s} {dt*1e3:8.2f} ms -> {n/dt:.3e} shots/s", flush=True)
    m = CompiledMeasurementSampler(prog, seed=1, noise=noise, **kwn)
    m.sample(n // 4, batch_size=1_000_000)
    t = time.perf_counter(); m.sample(n, batch_size=1_000_000); dt = time.perf_counter() - t
    print(f"noise={noise:6s} {'CompiledMeasurementSampler.sample':40s} {dt*1e3:8.2f} ms -> {n/dt:.3e} shots/s", flush=True)
