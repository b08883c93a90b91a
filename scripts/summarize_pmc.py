"""Summarise rocprofv3 counter_collection CSVs (one or more passes) for the sampling kernels.

usage: python scripts/summarize_pmc.py OUT.json CONFIG SHOTS DIR [DIR ...]
       TSIM_PMC_BATCHES_PER_LAUNCH=n: the run used fused first passes of n batches each (round 3: k_sample_lw_fast /
       k_sample_lw_multi with TSIM_AMD_FUSED_MAX=n; one k_sample_hw / k_sample4h_multi grid per group): the totals are
       then PER BATCH of SHOTS shots - first pass / n + hard-row grid / n.
Each DIR is a rocprofv3 -d output directory.  For every kernel whose name contains "k_sample" the
per-invocation MEDIAN of every counter is computed (bench.py's serial legs after the timed region launch the same
kernels with the padded 8-byte output word: an average would mix their 7.8 MB of writes into the 2.9 MB of the
timed launches - it did until v17: "WRITE_SIZE 4155 KB"); OUT.json holds the PER-LAUNCH totals (sum over
the kernels of one sampling launch: pattern-table pass + hard-row kernel + full kernel) at the top
level - the keys bench.py reads - and the per-kernel averages under "_per_kernel".

Deferred plan: one k_sample4h_multi grid serves the hard rows of GROUP launches (TSIM_AMD_DEFER_GROUP,
default 4), so a launch is k_sample_lw + 1/GROUP of that grid; the k_sample4h / k_sample4 invocations
in such a run belong to the first launches (before the plan has feedback) and are left out of the
total ("_weights" records what was summed).
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"tsimk::(k_\w+)", name)
    return m.group(1) if m else name


def main():
    out, config, shots, dirs = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        import os
        files = sorted(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
        for fn in files[-1:]:  # a re-used output directory keeps older runs: only the newest counts
            rows = [r for r in csv.DictReader(open(fn)) if "k_sample" in r["Kernel_Name"]]
            # fused first passes come in several grid sizes (the timed groups of TSIM_PMC_BATCHES_PER_LAUNCH batches; groups
            # of one batch from bench.py's serial / per-step context legs, which use the same kernel since round 3): only the
            # largest grid of a fused kernel is a timed launch
            biggest = {}
            for r in rows:
                k = short(r["Kernel_Name"])
                biggest[k] = max(biggest.get(k, 0), int(r.get("Grid_Size", 0) or 0))
            for r in rows:
                k = short(r["Kernel_Name"])
                g = int(r.get("Grid_Size", 0) or 0)
                if k in ("k_sample_lw_fast", "k_sample_lw_multi") and g != biggest[k]:
                    continue
                if k == "k_sample_hw" and 2 * g < biggest[k]:  # (its grid follows the list lengths: the groups of one batch are the small ones)
                    continue
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    import statistics
    per_kernel = {k: {c: statistics.median(v) for c, v in sorted(cs.items())} for k, cs in agg.items()}
    import os
    group = int(os.environ.get("TSIM_AMD_DEFER_GROUP", "4"))
    fused = int(os.environ.get("TSIM_PMC_BATCHES_PER_LAUNCH", "0"))
    if fused and any(k in per_kernel for k in ("k_sample_lw_fast", "k_sample_lw_multi")):
        weights = {k: 1.0 / fused for k in per_kernel if k in ("k_sample_lw_fast", "k_sample_lw_multi", "k_sample_hw", "k_sample4h_multi")}
    elif "k_sample4h_multi" in per_kernel:
        weights = {k: 1.0 for k in per_kernel if k.startswith("k_sample_lw")}  # k_sample_lw / k_sample_lw_reg
        weights["k_sample4h_multi"] = 1.0 / group
    else:
        weights = {k: 1.0 for k in per_kernel}
    total = collections.defaultdict(float)
    for k, cs in per_kernel.items():
        for c, v in cs.items():
            total[c] += v * weights.get(k, 0.0)
    res = dict(sorted(total.items()))
    res["_per_kernel"] = per_kernel
    res["_weights"] = weights
    res["_invocations"] = {k: {c: len(v) for c, v in cs.items()} for k, cs in agg.items()}
    res["_config"] = config
    res["_shots"] = shots
    res["_per"] = "batch of _shots shots" + (f" (fused launches of {fused} batches)" if fused else "")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
