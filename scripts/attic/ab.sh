#!/bin/bash
# usage: scripts/ab.sh lib1.so lib2.so ...   - serial first-pass time and the pipelined bench for each library, twice
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib"
  TSIM_AMD_LIB=$lib python scripts/lw_probe.py C2 64 0.02
  TSIM_AMD_LIB=$lib python bench.py --steps 200 --repeats 4 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
done; done
