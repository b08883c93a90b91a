#!/bin/bash
# A/B timing of k_sample_lw variants built into build_exp/lib_E*.so (run on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  export TSIM_AMD_LIB=$R/build_exp/lib_$e.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$e -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/ks_$e -name "*kernel_stats.csv" | head -1)
  echo "== $e"; grep -E "k_sample_lw|k_sample4" "$f" | cut -d, -f1-4
done
