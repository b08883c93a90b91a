#!/bin/bash
# usage: scripts/kstats.sh <tag> [bench args...]  - rocprofv3 kernel-trace stats of bench.py (run on the GPU box)
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$TAG -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-leg "$@" > $R/gpurun_out/ks_$TAG.json 2>/dev/null
f=$(find $R/gpurun_out/ks_$TAG -name "*kernel_stats.csv" | head -1)
cut -d, -f1-8 "$f" | head -12
tail -1 $R/gpurun_out/ks_$TAG.json | cut -c1-400
