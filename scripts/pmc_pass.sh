#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counters...>   (run on the GPU box through gpurun)
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
fn=glob.glob("$R/gpurun_out/pmc_$TAG/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fn)):
    if "k_sample" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(f"{k:28s} {sum(v)/len(v):.5g}")
PY
