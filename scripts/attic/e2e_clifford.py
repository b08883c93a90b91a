"""Clifford-only circuits end to end (CliffordCircuit(...).compile_detector_sampler().sample()): rotated surface code
memory, d rounds, 10^6 shots - host numpy path (the reference's _sample_direct) vs the device route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd.clifford import CliffordCircuit
from tsim_amd.circuits import rotated_surface_code_memory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for d in (3, 5):
    c = CliffordCircuit(rotated_surface_code_memory(d, d, after_clifford_depolarization=1e-3))
    for noise in ("host", "device"):
        s = c.compile_detector_sampler(seed=0, noise=noise)
        s.sample(n, bit_packed=True)
        for label, kw in (("bools", {}), ("bit_packed", {"bit_packed": True})):
            t = time.perf_counter(); s.sample(n, **kw); dt = time.perf_counter() - t
            print(f"d={d} num_f={s._channel_sampler.num_f} n_out={s._program.num_outputs} noise={noise:6s} {label:10s} {dt*1e3:8.2f} ms -> {n/dt:.3e} shots/s", flush=True)
    s = c.compile_detector_sampler(seed=0)
    t = time.perf_counter(); s._sample_direct(n); dt = time.perf_counter() - t
    print(f"d={d} host numpy path (_sample_direct)            {dt*1e3:8.2f} ms -> {n/dt:.3e} shots/s", flush=True)
