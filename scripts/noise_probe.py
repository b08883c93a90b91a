#!/usr/bin/env python3
"""Device noise sampler alone (k_noise_wave / k_noise_tile): us per 10^6 shots on the BASELINE noise models, resident buffers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsim_amd import backend, synth, prng
from tsim_amd.channels import ChannelSampler, error_probs

B = 1_000_000
for cn in (sys.argv[1:] or ["C2", "C3", "C5"]):
    prog, cfg = synth.config_program(cn)
    nf = cfg["num_f"]
    hp = backend.HipProgram(prog)
    cs = ChannelSampler([error_probs(cfg["p_bit"])] * nf, np.eye(nf, dtype=np.uint8), seed=1)
    dn = backend.DeviceNoiseSampler(hp, cs)
    WF = (nf + 63) // 64
    bufs = [hp.malloc(B * WF * 8) for _ in range(8)]
    key = prng.key(3)
    for b in bufs:
        dn.sample_into(b.ptr, B, key)
    hp.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(32):
            dn.sample_into(bufs[i % 8].ptr, B, (rep, i))
        hp.synchronize()
        ts.append((time.perf_counter() - t0) / 32)
    f = np.zeros((B, WF * 8), np.uint8)
    hp.d2h(f, bufs[0])
    bits = np.unpackbits(f, axis=1, bitorder="little")[:, :nf]
    print(f"{cn}: TSIM_AMD_TUNE={os.environ.get('TSIM_AMD_TUNE','')!r} noise kernel {sorted(ts)[2] * 1e6:.1f} us per 10^6 shots; mean fire rate {bits.mean():.5f} (model {cfg['p_bit']})", flush=True)
    hp.close()
