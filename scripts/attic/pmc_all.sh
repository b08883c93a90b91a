#!/bin/bash
# usage: scripts/pmc_all.sh <tag>   - the PMC passes behind profiles/latest_pmc.json (run on the GPU box).
# Counters are collected in separate passes with --kernel-trace only (no other trace domain).
# FETCH_SIZE and WRITE_SIZE go in separate passes: together they hang rocprofv3 on this workload.
R=$GRAFT_REPO_ROOT; TAG=$1; shift; EXTRA="$@"   # extra bench.py arguments, e.g. --config C3
cd /tmp && export TMPDIR=/tmp
LOG=$R/gpurun_out/pmc_$TAG.log; : > $LOG
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  echo "pass $i: $set  start $(date +%s)" >> $LOG
  timeout 60 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --repeats 1 $EXTRA >> $LOG 2>&1
  echo "pass $i rc=$? end $(date +%s)" >> $LOG
done
python $R/scripts/summarize_pmc.py $R/gpurun_out/pmc_$TAG.json C2 1000000 $R/gpurun_out/pmc_${TAG}_* > /dev/null
python - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc_$TAG.json"))
for k,v in d.items():
    if not k.startswith("_"): print(f"{k:24s} {v:.6g}")
for kn,cs in d["_per_kernel"].items():
    print(kn, {c: round(v,1) for c,v in cs.items() if c in ("SQ_INSTS_VALU","FETCH_SIZE","WRITE_SIZE","SQ_WAVES","SQ_WAVE_CYCLES")})
PY
