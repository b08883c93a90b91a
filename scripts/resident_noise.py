#!/usr/bin/env python3
"""The resident_device_noise leg of bench.py on its own (quick A/B): python scripts/resident_noise.py [C2|C3|C5]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-config-legs"],
                     capture_output=True, text=True)
d = json.loads(out.stdout.strip().splitlines()[-1])
r = d.get("resident_device_noise", {})
print(cfg, "value %.3e" % d["value"], {k: (round(v["shots_per_s"] / 1e10, 3), round(v["us_per_step"], 1), round(v["noise_kernel_alone_us_per_step"], 1)) for k, v in r.items() if isinstance(v, dict)}, r.get("error", ""))
