#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export TSIM_AMD_LIB=$R/build_exp/lib_E1.so
for cfg in "256 1000000" "1024 1000000" "512 1000000" "64 1000000" "256 4000000" "1024 4000000" "256 250000"; do
  set -- $cfg
  export TSIM_AMD_LW_BLOCK=$1
  rm -rf /tmp/ks_x
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_x -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --shots $2 > /dev/null 2>&1
  f=$(find /tmp/ks_x -name "*kernel_stats.csv" | head -1)
  echo "== blk=$1 shots=$2"; grep -E "k_sample_lw" "$f" | cut -d, -f1-4
done
