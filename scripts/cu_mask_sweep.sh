#!/bin/bash
# The hard-row grids of fused groups on compute units of their own (TSIM_AMD_HARD_CUS = n: two lanes confined to n CUs, the
# first-pass lanes to the other 256 - n; hipExtStreamCreateWithCUMask) against the default (the grid on the group's lane).
# Run on the GPU box; profiles/r04/cu_mask.txt.
R=$GRAFT_REPO_ROOT; cd $R
q() { scripts/bq.sh --no-config-legs "$@" | sed -e 's/enqueue_ms.*//'; }
for n in 0 8 16 32 64; do
  echo "== TSIM_AMD_HARD_CUS=$n"
  export TSIM_AMD_HARD_CUS=$n
  echo -n "C2 --steps 20  : "; q --steps 20 --warmup 5
  echo -n "C2 --steps 200 : "; q --steps 200
  echo -n "C2 --steps 1000: "; q --steps 1000 --repeats 3
  echo -n "C4 1e5 /step   : "; q --config C4 --shots 100000 --steps 100
  echo -n "C4 1e5, groups of 16 (TSIM_AMD_FUSED_MAX=16): "; TSIM_AMD_FUSED_MAX=16 q --config C4 --shots 100000 --steps 96
  echo -n "C3             : "; q --config C3 --steps 100
done
unset TSIM_AMD_HARD_CUS
