#!/bin/bash
# The measurements behind profiles/r04 (run on the GPU box through gpurun; results under gpurun_out/r04/).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
WHAT=" ${*:-all} "   # any of: bench shapes rocprof c5 transition long tests (default: all)
want() { [[ "$WHAT" == *" all "* || "$WHAT" == *" $1 "* ]]; }
q() { scripts/bq.sh --no-config-legs "$@" | sed -e 's/enqueue_ms.*//'; }
if want bench; then
  python bench.py > $O/bench.json 2> $O/bench.err
  python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench.err
  { echo "== bench.py --steps K (no spin-up, 64 resident f batches = 488 MB + 92 MB of outputs)"; for k in 5 10 20 50 200 1000; do q --steps $k; done;
    echo "== --spinup-ms 30"; for k in 20 200; do q --steps $k --spinup-ms 30; done;
    echo "== --nf 4 (round 3's working set: 32 MB of f, inside the Infinity Cache)"; for k in 20 200; do q --steps $k --nf 4; done; } > $O/steps_dependence.txt 2>&1
fi
if want shapes; then
  { for c in C3 C4 C5; do echo "== $c"; q --config $c --steps 100; done; echo "== C4, 1e5 shots per step"; q --config C4 --shots 100000 --steps 100;
    echo "== C5, round-2 path (TSIM_AMD_TUNE=wide_fused=0)"; TSIM_AMD_TUNE=wide_fused=0 q --config C5 --steps 100;
    echo "== C5 p_bit 0.005"; q --config C5 --steps 100 --p-bit 0.005; echo "== C5 p_bit 0.05"; q --config C5 --steps 100 --p-bit 0.05;
    echo "== C5, tables to weight 4 on request (TSIM_AMD_DEEP_TABLES=1)"; TSIM_AMD_DEEP_TABLES=1 q --config C5 --steps 100;
    echo "== C2 approx, live padding"; q --steps 200 --approx; q --steps 200 --live-padding;
    for p in 0.005 0.05 0.1 0.3; do echo "== C2 p_bit $p (400 untimed steps first: deeper tables, where the plan wants them, are built in the background)"; q --p-bit $p --steps 100 --warmup 400; done;
    echo "== C2, every row on the full kernel (TSIM_AMD_PATTERN_TABLES=0)"; TSIM_AMD_PATTERN_TABLES=0 q --steps 20; TSIM_AMD_PATTERN_TABLES=0 q --steps 20 --p-bit 0.3; } > $O/shapes.txt 2>&1
fi
if want rocprof; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/ks_driver /tmp/ks_default
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_driver -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-config-legs > $O/ks_driver.json 2>/dev/null
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_default -- python $R/bench.py --no-cpu-baseline --no-extra-legs --no-config-legs > $O/ks_default.json 2>/dev/null
  for t in driver default; do f=$(find /tmp/ks_$t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_${t}_cmd.csv; done
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/pmc_$i
    TSIM_BENCH_NO_CONTEXT=1 TSIM_AMD_TUNE=fused_max=4 timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra-legs --no-config-legs --repeats 1 > $O/pmc_$i.log 2>&1
  done
  cd $R
  TSIM_PMC_BATCHES_PER_LAUNCH=4 python scripts/summarize_pmc.py $O/pmc.json C2 1000000 /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 /tmp/pmc_4 /tmp/pmc_5 > /dev/null
  rm -f $O/pmc_[1-5].log
fi
if want c5; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/ks_c5
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_c5 -- python $R/bench.py --config C5 --steps 64 --no-cpu-baseline --no-extra-legs > $O/ks_c5.json 2>/dev/null
  f=$(find /tmp/ks_c5 -name "*kernel_stats.csv" | head -1); cp $f $O/c5_kernel_stats.csv
  cd $R
  PMC_TOP=2 scripts/pmc_kernel.sh c5 k_sample_wide --config C5 --no-config-legs > /dev/null 2>&1; cp $O/pmc_c5.txt $O/c5_pmc.txt   # (PMC_TOP: the 8-batch groups; the median mixes them with the 4-batch ones)
fi
if want transition; then
  { echo "C2, 10^6 shots per batch, 4 batches per call through tsim_sample_steps_device (scripts/dense_transition.py); ms per step of every call"
    for x in 0.05 0.1 0.3; do echo "== p_bit 0.02 -> $x -> 0.02"; TSIM_AMD_DEBUG=tables python scripts/dense_transition.py $x; done
    echo "== the jump to 0.3 with the overflow workers off (TSIM_AMD_TUNE=hard_overflow=0): the first dense call as in round 3"
    TSIM_AMD_TUNE=hard_overflow=0 python scripts/dense_transition.py 0.3 | sed -n 4,8p; } > $O/dense_transition.txt 2>&1
fi
if want long; then
  { echo "Long runs on one handle: scripts/bq.sh --no-config-legs --config X --steps 1500 (8 repetitions of 1500 steps of 10^6 shots), TSIM_AMD_DEBUG=tables, TSIM_AMD_TUNE=deep_after=4000000000."
    echo "After deep_after rows with too many hard / missed rows the next table depth is built in the background and swapped in; reps = ms per step of the 8 repetitions."
    for c in C3 C4 C5; do echo "== $c"; TSIM_AMD_TUNE=deep_after=4000000000 TSIM_AMD_DEBUG=tables scripts/bq.sh --no-config-legs --config $c --steps 1500 | sed -e "s/enqueue_ms.step [0-9.]* //"; grep "pattern tables" /tmp/bq.err | head -5; done; } > $O/long_runs.txt 2>&1
fi
if want tests; then
  python -m pytest tests -q -m gpu > $O/gpu_tests_full.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests_full.txt | tail -3 > $O/gpu_tests.txt
fi
ls -la $O
