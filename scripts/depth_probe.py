#!/usr/bin/env python3
"""Steady-state rate of a BASELINE configuration against the depth of its pattern tables (pinned at finalize):
    python scripts/depth_probe.py [C2,C3,C4] [5,6,7] [steps per region]
The deeper the table, the fewer hard rows behind the first pass (C2: ~60 per 10^6 shots at weight 5, ~5 at 6)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.shape_map import measure
from tsim_amd import backend, synth

cfgs = (sys.argv[1] if len(sys.argv) > 1 else "C2,C3,C4").split(",")
depths = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "5,6,7").split(",")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for cn in cfgs:
    prog, cfg = synth.config_program(cn)
    for d in depths:
        r = measure(backend, prog, cfg["num_f"], cfg["p_bit"], 1_000_000, steps, pattern_tables=d)
        print(f"{cn} depth {r['depth']} tables {r['table_mb']:.0f} MB  {r['rate']:.3e} shots/s  {r['us']:.2f} us/step  paths {r['paths']}", flush=True)
