"""Repeat the bit-packed device pipeline call to separate steady state from first-call effects."""
import sys, time, warnings
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
probs = [error_probs(cfg["p_bit"])] * cfg["num_f"]
T = np.eye(cfg["num_f"], dtype=np.uint8)
s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=1, noise="device")
shots, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, 1_000_000
for packed in (True, False):
    for i in range(6):
        t0 = time.perf_counter()
        out = s.sample(shots, batch_size=batch, bit_packed=packed, append_observables=True)
        dt = time.perf_counter() - t0
        print(f"packed={packed} call {i}: {dt*1e3:7.2f} ms  {shots/dt:.3e} shots/s  {out.shape}")
