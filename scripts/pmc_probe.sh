#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, env..., counters
  name=$1; shift
  s=$(date +%s)
  env "$@" timeout 45 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pp_$name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-leg > /tmp/pp_$name.log 2>&1
  echo "$name ctr=[$CTR] rc=$? $(( $(date +%s) - s ))s"
}
CTR="FETCH_SIZE" run fetch_default A=1
CTR="WRITE_SIZE" run write_default A=1
CTR="FETCH_SIZE" run fetch_nohard TSIM_AMD_HARD_KERNEL=0
CTR="FETCH_SIZE" run fetch_notables TSIM_AMD_PATTERN_TABLES=0
