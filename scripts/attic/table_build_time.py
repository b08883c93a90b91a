"""Wall time of the pattern-table builds (TSIM_TABLE_TIMING=1 makes the library print one line per build):
the one at program load and the on-demand deepening (TSIM_AMD_DEEP_TABLES=1, triggered here by a dense batch)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, TSIM_TABLE_TIMING="1", TSIM_AMD_DEEP_TABLES="1", TSIM_BENCH_NO_CONTEXT="1")
for args in (["--config", "C2"], ["--config", "C2", "--p-bit", "0.05"], ["--config", "C3"], ["--config", "C4"], ["--config", "C5"]):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", *args],
                       env=env, capture_output=True, text=True)
    lines = [l for l in r.stderr.splitlines() if "pattern tables" in l]
    print(" ".join(args), "->", "; ".join(l.split("tables: ")[1] for l in lines) or r.stderr[-300:])
