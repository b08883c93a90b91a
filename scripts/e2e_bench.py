"""End-to-end `CompiledDetectorSampler.sample()` throughput (host buffers in/out, PCIe included).

C2 program, noise model: 64 independent one-bit channels (p = 0.02) with identity error_transform,
i.e. the same per-bit fire rate as bench.py's synthetic f.  Not the `value` of bench.py.
"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import warnings
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler

warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
p = float(sys.argv[1]) if len(sys.argv) > 1 else cfg["p_bit"]
probs = [error_probs(p)] * cfg["num_f"]
T = np.eye(cfg["num_f"], dtype=np.uint8)
shots, batch = 4_000_000, 1_000_000
for mode in ("host", "device"):
    s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=1, noise=mode)
    s.sample(shots, batch_size=batch)  # warm-up with the same shape (program upload, buffer allocations)
    t0 = time.perf_counter()
    out = s.sample(shots, batch_size=batch, append_observables=True)
    dt = time.perf_counter() - t0
    print(f"noise={mode:6s} p={p}: {shots/dt:.3e} shots/s end to end ({dt*1e3/ (shots/1e6):.1f} ms per 1e6 shots), out {out.shape} {out.dtype}")
    if mode == "host":
        t0 = time.perf_counter(); f = s._channel_sampler.sample(batch); dt = time.perf_counter() - t0
        print(f"   host ChannelSampler.sample alone (unpacked rows): {batch/dt:.3e} shots/s")
        t0 = time.perf_counter(); f = s._channel_sampler.sample_packed(batch); dt = time.perf_counter() - t0
        print(f"   host ChannelSampler.sample_packed alone:          {batch/dt:.3e} shots/s")
s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=1, noise="device")
s.sample(shots, batch_size=batch, bit_packed=True, append_observables=True)
t0 = time.perf_counter()
out = s.sample(shots, batch_size=batch, bit_packed=True, append_observables=True)
dt = time.perf_counter() - t0
print(f"noise=device bit_packed=True p={p}: {shots/dt:.3e} shots/s end to end ({dt*1e3/(shots/1e6):.2f} ms per 1e6 shots), out {out.shape} {out.dtype}")
