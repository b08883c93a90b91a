#!/bin/bash
# usage: scripts/pmc_shape.sh <tag> <kernel substring> <shape class> [TSIM_AMD_TUNE value]
# rocprofv3 counter passes (each with --kernel-trace only) over scripts/shape_map.py for ONE shape class; prints the mean of the
# largest invocations of the named kernel per counter (scripts/pmc_kernel.py) -> gpurun_out/r05/pmc_<tag>.txt
R=$GRAFT_REPO_ROOT; TAG=$1; KERN=$2; CLS=$3; TUNE=$4
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmcs_${TAG}_$i
  TSIM_AMD_TUNE=$TUNE timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcs_${TAG}_$i -- python $R/scripts/shape_map.py --no-base --classes $CLS --steps 16 > /tmp/pmcs_${TAG}_$i.log 2>&1
done
PMC_TOP=4 python $R/scripts/pmc_kernel.py "$KERN" /tmp/pmcs_${TAG}_* > $R/gpurun_out/r05/pmc_$TAG.txt 2>&1
cat $R/gpurun_out/r05/pmc_$TAG.txt
