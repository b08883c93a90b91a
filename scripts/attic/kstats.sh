#!/bin/bash
# usage: scripts/kstats.sh <tag> [bench args...]  - rocprofv3 kernel-trace stats of bench.py (run on the GPU box)
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$TAG -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --repeats 1 "$@" > $R/gpurun_out/ks_$TAG.json 2>/dev/null
f=$(find $R/gpurun_out/ks_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
tail -1 $R/gpurun_out/ks_$TAG.json | cut -c1-300
