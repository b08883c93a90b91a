"""Bandwidth of the data-format kernels (k_pack_bits / k_unpack_bits): HBM-bound rows of the path."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth

prog, cfg = synth.config_program("C2")
hp = backend.HipProgram(prog)
B = 8_000_000
for nbits in (64, 20, 320):
    wq = (nbits + 63) // 64
    a = (np.random.default_rng(0).random((B, nbits)) < 0.05).astype(np.uint8)
    d_in, d_p, d_out = hp.malloc(a.nbytes), hp.malloc(B * wq * 8), hp.malloc(a.nbytes)
    hp.h2d(d_in, a)
    for name, fn, nbytes in (
        ("pack", lambda: hp.pack_bits_device(d_in.ptr, B, nbits, d_p.ptr), B * (nbits + wq * 8)),
        ("unpack", lambda: hp.unpack_bits_device(d_p.ptr, B, nbits, d_out.ptr), B * (nbits + wq * 8)),
    ):
        fn(); hp.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        hp.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name:6s} nbits={nbits:4d}: {dt*1e3:7.3f} ms  {nbytes/dt/1e9:8.1f} GB/s  ({nbytes/dt/8e12*100:.1f}% of 8 TB/s)")
    for b in (d_in, d_p, d_out): b.free()
