#!/usr/bin/env python3
"""Where a fresh handle's time goes: TSIM_AMD_DEBUG=finalize marks of tsim_program_finalize + the Python side of HipProgram().
    TSIM_AMD_DEBUG=finalize python scripts/cold_probe.py C4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tsim_amd import backend, synth

for name in sys.argv[1:] or ["C4"]:
    program, cfg = synth.config_program(name)
    helper = backend.HipProgram(program)
    helper.synchronize()
    for rep in range(3):
        print(f"--- {name} handle {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        hp = backend.HipProgram(program)
        t1 = time.perf_counter()
        print(f"--- {name} handle {rep}: {(t1 - t0) * 1e3:.2f} ms", file=sys.stderr, flush=True)
        hp.close()
    helper.close()
