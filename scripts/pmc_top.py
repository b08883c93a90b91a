"""Per-launch HBM traffic and instruction counters of the timed launches of bench.py's headline leg.

usage: python scripts/pmc_top.py OUT.json CONFIG SHOTS BATCHES_PER_LAUNCH FIRST_PASS_KERNEL[,HARD_ROW_KERNEL...] DIR [DIR ...]

Each DIR is a rocprofv3 -d directory of ONE counter pass (`--pmc <set> --kernel-trace`, nothing else) over the same bench.py
command.  A fused launch puts one chip-full of blocks on the GPU whatever the number of batches in its group, so the grid does
not tell the timed groups (BATCHES_PER_LAUNCH batches) from the smaller ones bench.py also makes (initialisation in calls of
four, the last group of a region): per counter the TWO LARGEST invocations of a kernel are taken - the rule of
profiles/r04/c2_first_pass_pmc.txt - and their mean is that kernel's PER-LAUNCH figure.  OUT.json: `_per_launch` holds those
per kernel; the top level holds the sum over the named kernels divided by BATCHES_PER_LAUNCH = PER BATCH of SHOTS shots, the
unit bench.py scales by its own batches per launch (`roofline.traffic` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch;
VERDICT r04 item 4: never below the algorithmic bytes - bench.py refuses a figure that is).
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, config, shots, bpl = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    kernels, dirs = sys.argv[5].split(","), sys.argv[6:]
    vals = {k: collections.defaultdict(list) for k in kernels}
    for d in dirs:
        files = sorted(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
        for fn in files[-1:]:
            for r in csv.DictReader(open(fn)):
                for k in kernels:
                    if k in r["Kernel_Name"]:
                        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    per_launch = {k: {c: sum(sorted(v)[-2:]) / len(sorted(v)[-2:]) for c, v in sorted(cs.items())} for k, cs in vals.items() if cs}
    total = collections.defaultdict(float)
    for k, cs in per_launch.items():
        for c, v in cs.items():
            total[c] += v / bpl
    res = dict(sorted(total.items()))
    res["_per_launch"] = per_launch
    res["_invocations"] = {k: {c: len(v) for c, v in cs.items()} for k, cs in vals.items()}
    res["_batches_per_launch"] = bpl
    res["_config"] = config
    res["_shots"] = shots
    res["_per"] = f"batch of _shots shots: (mean of the two largest invocations per counter and kernel, summed over {kernels}) / {bpl} batches per launch"
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        res["_traffic_bytes_per_batch"] = (2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("_inv")}, indent=1))


if __name__ == "__main__":
    main()
