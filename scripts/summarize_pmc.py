"""Summarise rocprofv3 counter_collection CSVs (one or more passes) for the sampling kernel.

usage: python scripts/summarize_pmc.py OUT.json DIR [DIR ...]
Each DIR is a rocprofv3 -d output directory; per-launch averages of every counter found for
kernels whose name contains "k_sample" are written to OUT.json (and printed).
"""
import collections
import csv
import glob
import json
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(list)
    for d in dirs:
        for fn in glob.glob(f"{d}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(fn)):
                if "k_sample" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    kern = r["Kernel_Name"]
    res = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
    res["_kernel"] = kern
    res["_launches_per_counter"] = {k: len(v) for k, v in agg.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
