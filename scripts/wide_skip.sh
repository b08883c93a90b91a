#!/bin/bash
# usage: scripts/wide_skip.sh build   (here: diagnostic variants of libtsim_hip.so that leave parts of k_sample_wide out)
#        scripts/wide_skip.sh run     (on the GPU box: C5, 8 batches per call, the time of each variant)
MASKS="1 2 4 8 16 32 64 65 127"
if [ "$1" = build ]; then
  for m in $MASKS; do scripts/build_variant.sh WORK scripts/_ab_wskip$m.so -DTSIMK_WIDE_SKIP=$m > /dev/null 2>&1 & done; wait; ls scripts/_ab_wskip*.so
else
  echo -n "everything                        : "; python scripts/wide_trace.py C5 8 | tail -1
  for m in $MASKS; do echo -n "TSIMK_WIDE_SKIP=$m : "; TSIM_AMD_ALLOW_STALE=1 TSIM_AMD_LIB=scripts/_ab_wskip$m.so python scripts/wide_trace.py C5 8 2>&1 | tail -1; done
  echo "(1 = dense passes without their levels, 2 = no atomics, 4 = thresholds of pattern 0 only, 8 = no Threefry, 16 = no set-bit walk (every row weight 0), 32 = no row stores, 64 = nothing queued (no dense passes), 65 / 127 = combinations)"
fi
