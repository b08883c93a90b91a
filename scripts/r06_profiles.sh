#!/bin/bash
# The measurements behind profiles/r06 (run on the GPU box through gpurun; results under gpurun_out/r06/).
# usage: scripts/r06_profiles.sh [bench] [approx] [rocprof] [pmc] [noise] [shape] [classes] [cold] [seam] [legs] [tests] [fuzz]   (default: all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
WHAT=" ${*:-all} "
want() { [[ "$WHAT" == *" all "* || "$WHAT" == *" $1 "* ]]; }
q() { scripts/bq.sh --no-config-legs "$@" | sed -e 's/enqueue_ms.*//'; }
if want bench; then
  python bench.py > $O/bench.json 2> $O/bench.err
  python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench.err
  { echo "== bench.py --steps K (no spin-up, 64 resident f batches)"; for k in 5 10 20 50 200 1000; do q --steps $k; done; } > $O/steps_dependence.txt 2>&1
fi
if want approx; then  # the branch the real circuits take (has_approximate_floatfactors), as headline, at steady state
  python bench.py --approx --no-extra-legs > $O/bench_approx.json 2>> $O/bench.err
  python bench.py --approx --steps 20 --warmup 5 --no-extra-legs > $O/bench_approx_steps20.json 2>> $O/bench.err
fi
if want rocprof; then
  cd /tmp && export TMPDIR=/tmp
  for t in "driver --steps 20 --warmup 5" "default" "approx --approx --steps 20 --warmup 5" "C3 --config C3 --steps 64" "C4 --config C4 --shots 100000 --steps 64" "C5 --config C5 --steps 64"; do
    set -- $t; tag=$1; shift
    rm -rf /tmp/ks_$tag
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python $R/bench.py "$@" --no-cpu-baseline --no-extra-legs --no-config-legs > $O/ks_$tag.json 2>/dev/null
    f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_${tag}_cmd.csv
  done
  cd $R
fi
if want pmc; then
  cd /tmp && export TMPDIR=/tmp
  i=0; dirs=""
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/pmc6_$i
    TSIM_BENCH_NO_CONTEXT=1 timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc6_$i -- python $R/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra-legs --no-config-legs --repeats 2 > $O/pmc_$i.log 2>&1
    dirs="$dirs /tmp/pmc6_$i"
  done
  cd $R
  python scripts/pmc_top.py $O/pmc.json C2 1000000 8 k_sample_lw_fast,k_sample_hw $dirs > $O/pmc_summary.txt 2>&1
  rm -f $O/pmc_[1-5].log
  i=0; dirs=""
  cd /tmp
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"; do
    i=$((i+1)); rm -rf /tmp/pmc6c_$i
    TSIM_BENCH_NO_CONTEXT=1 timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc6c_$i -- python $R/bench.py --config C5 --steps 32 --warmup 8 --no-cpu-baseline --no-extra-legs --no-config-legs --repeats 2 > /dev/null 2>&1
    dirs="$dirs /tmp/pmc6c_$i"
  done
  cd $R
  python scripts/pmc_top.py $O/c5_pmc.json C5 1000000 8 k_sample_wide $dirs > $O/c5_pmc.txt 2>&1
fi
if want noise; then  # the device noise sampler alone - k_noise_wave (round 6) against k_noise_tile (round 3)
  cd /tmp && export TMPDIR=/tmp
  { echo "== scripts/noise_probe.py: us per 10^6 shots, resident buffers"; python $R/scripts/noise_probe.py; TSIM_AMD_TUNE=noise_wave=0 python $R/scripts/noise_probe.py; } > $O/noise_kernels.txt 2>&1
  for v in wave tile; do
    i=0; dirs=""
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
      i=$((i+1)); rm -rf /tmp/pmcn_${v}_$i
      TSIM_AMD_TUNE=noise_wave=$([ $v = wave ] && echo 1 || echo 0) timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcn_${v}_$i -- python $R/scripts/noise_probe.py C2 > /dev/null 2>&1
      dirs="$dirs /tmp/pmcn_${v}_$i"
    done
    python $R/scripts/pmc_top.py $O/noise_pmc_$v.json C2 1000000 1 k_noise_$v $dirs > $O/noise_pmc_$v.txt 2>&1
  done
  cd $R
fi
if want shape; then
  python scripts/shape_map.py --check --out $O/shape_map.txt > /dev/null 2>&1
fi
if want classes; then  # kernel stats of the classes round 6 added (which kernel serves them, how long a launch takes)
  cd /tmp && export TMPDIR=/tmp
  for c in n13 n16 n24 n40 20narrow w12 9wide F600 F140; do
    rm -rf /tmp/ksc_$c
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksc_$c -- python $R/scripts/shape_map.py --classes $c --no-base > /tmp/ksc_$c.txt 2>/dev/null
    f=$(find /tmp/ksc_$c -name "*kernel_stats.csv" | head -1)
    { grep "^$c " /tmp/ksc_$c.txt | sed 's/^/# /'; head -12 $f; } > $O/kernel_stats_class_$c.csv
  done
  cd $R
fi
if want cold; then
  python scripts/time_to_n.py > $O/time_to_n.txt 2>&1
  # where a fresh handle's time goes (marks of tsim_program_finalize; the third handle of each configuration is the settled one)
  TSIM_AMD_DEBUG=finalize python scripts/cold_probe.py C4 C2 2>&1 | sed -e 's/ (image.*//' > $O/cold_breakdown.txt
fi
if want seam; then  # the one-batch API (backend.sample_program's seam) on programs whose tables only k_sample_gen reads: before / after
  { for t in "gen=0" ""; do echo "== TSIM_AMD_TUNE='$t'"; TSIM_AMD_TUNE=$t python scripts/one_batch_probe.py n16 n24 n40 F70 2>&1 | grep "one-batch"; done; } > $O/one_batch_api.txt 2>&1
  scripts/pmc_shape.sh gen20 k_sample_gen 20narrow > /dev/null 2>&1; cp $R/gpurun_out/r05/pmc_gen20.txt $O/pmc_gen_20narrow.txt
fi
if want legs; then
  { for c in C3 C4 C5; do echo "== $c"; q --config $c --steps 100; done; echo "== C4, 1e5 shots per step"; q --config C4 --shots 100000 --steps 100;
    echo "== C2, every row on the full kernel (TSIM_AMD_PATTERN_TABLES=0)"; TSIM_AMD_PATTERN_TABLES=0 q --steps 20; TSIM_AMD_PATTERN_TABLES=0 q --steps 20 --p-bit 0.3; } > $O/shapes.txt 2>&1
fi
if want tests; then
  python -m pytest tests -q -m gpu > $O/gpu_tests_full.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests_full.txt | tail -3 > $O/gpu_tests.txt
fi
if want fuzz; then
  { python scripts/fuzz_steps.py 6 60 0; python scripts/fuzz_steps.py 8 40 1000; } > $O/fuzz_soak.txt 2>&1
fi
ls -la $O
