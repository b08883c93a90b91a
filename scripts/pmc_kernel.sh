#!/bin/bash
# usage: scripts/pmc_kernel.sh <tag> <kernel substring> <bench.py arguments...>
# rocprofv3 counter passes (each with --kernel-trace only) over a short bench.py run; prints the per-invocation MEDIAN of
# every counter for the invocations of the named kernel with the largest grid, and its kernel-trace duration.
R=$GRAFT_REPO_ROOT; TAG=$1; KERN=$2; shift; shift
mkdir -p $R/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  TSIM_BENCH_NO_CONTEXT=1 timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -- python $R/bench.py --steps 16 --warmup 4 --spinup-ms 0 --no-cpu-baseline --no-extra-legs --repeats 1 "$@" > /tmp/pmc_${TAG}_$i.log 2>&1
done
python $R/scripts/pmc_kernel.py "$KERN" /tmp/pmc_${TAG}_* > $R/gpurun_out/r04/pmc_$TAG.txt 2>&1
cat $R/gpurun_out/r04/pmc_$TAG.txt
