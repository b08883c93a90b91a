#!/bin/bash
# The measurements behind profiles/r05 (run on the GPU box through gpurun; results under gpurun_out/r05/).
# usage: scripts/r05_profiles.sh [bench] [legs] [rocprof] [pmc] [shape] [cold] [tests]   (default: all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
WHAT=" ${*:-all} "
want() { [[ "$WHAT" == *" all "* || "$WHAT" == *" $1 "* ]]; }
q() { scripts/bq.sh --no-config-legs "$@" | sed -e 's/enqueue_ms.*//'; }
if want bench; then
  python bench.py > $O/bench.json 2> $O/bench.err
  python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench.err
  { echo "== bench.py --steps K (no spin-up, 64 resident f batches)"; for k in 5 10 20 50 200 1000; do q --steps $k; done; } > $O/steps_dependence.txt 2>&1
fi
if want rocprof; then
  cd /tmp && export TMPDIR=/tmp
  # kernel stats of the driver's command, of the default command, and of the C3 / C4 / C5 legs run as headline
  for t in "driver --steps 20 --warmup 5" "default" "C3 --config C3 --steps 64" "C4 --config C4 --shots 100000 --steps 64" "C5 --config C5 --steps 64"; do
    set -- $t; tag=$1; shift
    rm -rf /tmp/ks_$tag
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python $R/bench.py "$@" --no-cpu-baseline --no-extra-legs --no-config-legs > $O/ks_$tag.json 2>/dev/null
    f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_${tag}_cmd.csv
  done
  cd $R
fi
if want pmc; then
  cd /tmp && export TMPDIR=/tmp
  # HBM traffic and instruction counters of the DRIVER's command, one counter set per pass, nothing but --kernel-trace beside --pmc
  i=0; dirs=""
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/pmc5_$i
    TSIM_BENCH_NO_CONTEXT=1 timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc5_$i -- python $R/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra-legs --no-config-legs --repeats 2 > $O/pmc_$i.log 2>&1
    dirs="$dirs /tmp/pmc5_$i"
  done
  cd $R
  python scripts/pmc_top.py $O/pmc.json C2 1000000 8 k_sample_lw_fast,k_sample_hw $dirs > $O/pmc_summary.txt 2>&1
  rm -f $O/pmc_[1-5].log
  # the same for C5's kernel
  i=0; dirs=""
  cd /tmp
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"; do
    i=$((i+1)); rm -rf /tmp/pmc5c_$i
    TSIM_BENCH_NO_CONTEXT=1 timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc5c_$i -- python $R/bench.py --config C5 --steps 32 --warmup 8 --no-cpu-baseline --no-extra-legs --no-config-legs --repeats 2 > /dev/null 2>&1
    dirs="$dirs /tmp/pmc5c_$i"
  done
  cd $R
  python scripts/pmc_top.py $O/c5_pmc.json C5 1000000 8 k_sample_wide $dirs > $O/c5_pmc.txt 2>&1
fi
if want shape; then
  python scripts/shape_map.py --check --out $O/shape_map.txt > /dev/null 2>&1
fi
if want cold; then
  python scripts/time_to_n.py > $O/time_to_n.txt 2>&1
  if [ -x scripts/microbench/hip_setup_cost.bin ] || hipcc --offload-arch=gfx950 -O2 scripts/microbench/hip_setup_cost.hip -o scripts/microbench/hip_setup_cost.bin 2>/dev/null; then scripts/microbench/hip_setup_cost.bin > $O/hip_setup_cost.txt 2>&1; fi
fi
if want legs; then
  { for c in C3 C4 C5; do echo "== $c"; q --config $c --steps 100; done; echo "== C4, 1e5 shots per step"; q --config C4 --shots 100000 --steps 100;
    echo "== C2, every row on the full kernel (TSIM_AMD_PATTERN_TABLES=0)"; TSIM_AMD_PATTERN_TABLES=0 q --steps 20; TSIM_AMD_PATTERN_TABLES=0 q --steps 20 --p-bit 0.3; } > $O/shapes.txt 2>&1
fi
if want tests; then
  python -m pytest tests -q -m gpu > $O/gpu_tests_full.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests_full.txt | tail -3 > $O/gpu_tests.txt
fi
ls -la $O
