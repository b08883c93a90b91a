"""Calibration of rocprofv3's WRITE_SIZE on known byte counts: k_compact_rows writes exactly B * rb bytes (dword stores),
k_unpack_bits B * n bytes (wide stores), hipMemset n bytes.  Run under rocprofv3 --pmc WRITE_SIZE --kernel-trace."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tsim_amd import backend, synth
prog, cfg = synth.config_program("C2")
hp = backend.HipProgram(prog)
B = 1_000_000
d_in = hp.malloc(B * 8); d_out = hp.malloc(B * 8 + 64)
hp.h2d(d_in, np.random.default_rng(1).integers(0, 255, B * 8, dtype=np.uint8))
for nbits in (20, 24, 32, 45, 64):
    for _ in range(5):
        hp.compact_rows_device(d_in.ptr, B, nbits, d_out.ptr)
    hp.synchronize()
    print("compact rows", nbits, "bits:", B * ((nbits + 7) // 8), "bytes written per call")
