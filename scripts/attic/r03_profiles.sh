#!/bin/bash
# The measurements behind profiles/r03 (run on the GPU box through gpurun; results under gpurun_out/r03/).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
WHAT=" ${*:-all} "   # any of: micro bench e2e breakdown rocprof tests (default: all)
want() { [[ "$WHAT" == *" all "* || "$WHAT" == *" $1 "* ]]; }
if want micro; then
  ./scripts/microbench/valu_table.bin > $O/valu_table.txt 2>&1
  ./scripts/microbench/threefry_block.bin > $O/threefry_block.txt 2>&1
  ./scripts/microbench/salu_mix.bin > $O/salu_mix.txt 2>&1
fi
if want bench; then
  python bench.py > $O/bench.json 2> $O/bench.err
  python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench.err
  python bench.py --approx --no-cpu-baseline --no-extra-legs > $O/bench_approx.json 2>> $O/bench.err
  python bench.py --approx --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench_approx_steps20.json 2>> $O/bench.err
  { echo "== default (30 ms of untimed steps before every timed region: sustained clocks)"; for k in 5 10 20 50 200 1000; do scripts/bq.sh --steps $k; done;
    echo "== --spinup-ms 0 (the repetitions of one process get faster until ~25 ms of work have gone by)"; for k in 5 10 20 50 200 1000; do scripts/bq.sh --steps $k --spinup-ms 0; done; } > $O/steps_dependence.txt 2>&1
  { for c in C3 C4 C5; do echo "== $c"; scripts/bq.sh --config $c --steps 100; done; echo "== C4, 1e5 shots per step"; scripts/bq.sh --config C4 --shots 100000 --steps 100;
    echo "== C2 per-step API (rounds 1-2)"; TSIM_BENCH_PER_STEP=1 scripts/bq.sh --steps 200; TSIM_BENCH_PER_STEP=1 scripts/bq.sh --steps 20;
    echo "== C2 generic fused pass (TSIM_AMD_LW_FAST=0)"; TSIM_AMD_LW_FAST=0 scripts/bq.sh --steps 200;
    echo "== C2 hard rows on k_sample4h_multi (TSIM_AMD_HARD_WAVE=0)"; TSIM_AMD_HARD_WAVE=0 scripts/bq.sh --steps 200; TSIM_AMD_HARD_WAVE=0 scripts/bq.sh --steps 20; TSIM_AMD_HARD_WAVE=0 scripts/bq.sh --steps 200 --approx;
    echo "== C2 hard-row grids on the batch lane (TSIM_AMD_HARD_INLINE_ROWS=0: rounds 1-2)"; TSIM_AMD_HARD_INLINE_ROWS=0 scripts/bq.sh --steps 200; TSIM_AMD_HARD_INLINE_ROWS=0 scripts/bq.sh --steps 20;
    echo "== C4, 1e5 shots per step, batch lane"; TSIM_AMD_HARD_INLINE_ROWS=0 scripts/bq.sh --config C4 --shots 100000 --steps 100;
    echo "== C3 block-per-row kernel forced (TSIM_AMD_HARD_WAVE_ROWS=100000)"; TSIM_AMD_HARD_WAVE_ROWS=100000 scripts/bq.sh --config C3 --steps 100;
    echo "== deeper pattern tables on demand (TSIM_AMD_DEEP_TABLES=1: one more weight when the hard rows are too many for the block-per-row kernel; 60-200 ms of table build once per handle; default: only after 2e10 rows in that state): C3, C4, C2 at p_bit 0.05"; TSIM_AMD_DEEP_TABLES=1 scripts/bq.sh --config C3 --steps 100; TSIM_AMD_DEEP_TABLES=1 scripts/bq.sh --config C4 --steps 100; TSIM_AMD_DEEP_TABLES=1 scripts/bq.sh --p-bit 0.05 --steps 100;
    echo "== C2 approx, live padding"; scripts/bq.sh --steps 200 --approx; scripts/bq.sh --steps 200 --live-padding;
    for p in 0.005 0.05 0.1 0.3; do echo "== C2 p_bit $p"; scripts/bq.sh --p-bit $p --steps 100; done; } > $O/shapes.txt 2>&1
fi
if want e2e; then
  { python scripts/e2e_breakdown.py p_bit; python scripts/e2e_breakdown.py p1e-3; for n in 4000000 16000000 64000000; do python scripts/e2e_profile2.py $n | head -1; done;
    TSIM_AMD_NOISE_ATOMIC=1 python scripts/e2e_profile2.py 16000000 | head -1; python scripts/e2e_host.py 8000000; TSIM_PCG_TIMING=1 python scripts/e2e_host.py 2000000 2>&1 | tail -12;
    echo "== serial stream (TSIM_PCG_SERIAL=1: round 2's sampler)"; TSIM_PCG_SERIAL=1 python scripts/e2e_host.py 8000000;
    g++ -O3 -std=c++17 -msse4.1 -ffp-contract=off -pthread scripts/microbench/pcg_probe.cpp -o /tmp/pcg_probe && /tmp/pcg_probe; lscpu | grep -E "Model name|^CPU\(s\)"; cat /sys/fs/cgroup/cpu.max; } > $O/e2e.txt 2>&1
fi
if want breakdown; then
  # diagnostic builds that leave parts of the first pass out (wrong results; built here when absent: hipcc is on the box)
  for m in 1 2 4 16 32 64 112 113; do [ -f scripts/_ab_skip$m.so ] || scripts/build_variant.sh WORK scripts/_ab_skip$m.so -DTSIMK_LWM_SKIP=$m > /dev/null 2>&1; done
  { echo "fused first pass alone (scripts/lwm_probe.py: 8 batches per launch, serial launches, HIP events), parts left out (diagnostic builds, wrong results):";
    echo -n "everything                         : "; python scripts/lwm_probe.py C2 | tail -1
    for m in 1 2 4 16 32 64 112 113; do echo -n "TSIMK_LWM_SKIP=$m : "; TSIM_AMD_ALLOW_STALE=1 TSIM_AMD_LIB=scripts/_ab_skip$m.so python scripts/lwm_probe.py C2 2>&1 | tail -1; done;
    echo "(1 = no Threefry blocks, 2 = no direct outputs, 4 = no rank loop, 16 = no stores, 32 = no threshold reads, 64 = no f loads, 112 = no memory at all, 113 = no memory and no Threefry)";
    echo -n "generic fused pass (TSIM_AMD_LW_FAST=0): "; TSIM_AMD_LW_FAST=0 python scripts/lwm_probe.py C2 | tail -1; echo -n "C3: "; python scripts/lwm_probe.py C3 | tail -1; } > $O/first_pass_breakdown.txt 2>&1
fi
if want rocprof; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/ks_pipe $O/ks_serial
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_pipe -- python $R/bench.py --steps 20 --warmup 5 --repeats 4 --spinup-ms 0 --no-cpu-baseline --no-extra-legs > $O/ks_pipe.json 2>/dev/null
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_serial -- python $R/scripts/lwm_probe.py C2 > $O/ks_serial.txt 2>/dev/null
  for t in pipe serial; do f=$(find $O/ks_$t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$t.csv; done
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    TSIM_BENCH_NO_CONTEXT=1 TSIM_AMD_FUSED_MAX=4 timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -- python $R/bench.py --steps 16 --warmup 4 --spinup-ms 0 --no-cpu-baseline --no-extra-legs --repeats 1 > $O/pmc_$i.log 2>&1
  done
  cd $R
  TSIM_PMC_BATCHES_PER_LAUNCH=4 python scripts/summarize_pmc.py $O/pmc.json C2 1000000 $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 $O/pmc_5 > /dev/null
  rm -rf $O/ks_pipe $O/ks_serial $O/pmc_[1-5]
fi
if want tests; then
  python -m pytest tests -q -m gpu > $O/gpu_tests_full.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests_full.txt | tail -3 > $O/gpu_tests.txt
fi
ls -la $O
