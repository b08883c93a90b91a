#!/bin/bash
# usage: scripts/build_sample_variant.sh <out.so> [extra hipcc flags]
# A/B variants of the kernels in tsim_sample.hip only (k_sample_wide, the first passes, k_sample4*): compiles that one
# translation unit of the WORKING TREE with the extra flags and links it with the library's other objects
# (tsim_amd/_build, built by tsim_amd.build from the same tree).  TSIM_AMD_ALLOW_STALE=1 TSIM_AMD_LIB=<out.so> runs it.
set -e
OUT=$(realpath -m $1); shift
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
TU=${TSIM_VARIANT_TU:-tsim_sample}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c $R/tsim_amd/csrc/$TU.hip -o $T/$TU.hip.o 2>/dev/null
OBJS=$(ls $R/tsim_amd/_build/*.o | grep -v "/$TU.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/$TU.hip.o -o $OUT -lrccl
rm -rf $T
echo built $OUT
