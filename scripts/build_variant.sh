#!/bin/bash
# usage: scripts/build_variant.sh <git-ref|WORK> <out.so> [extra hipcc flags]
# Builds libtsim_hip from the sources of a git ref (or of the working tree) into <out.so> - for A/B runs on one GPU box:
#   TSIM_AMD_LIB=scripts/_ab_old.so python scripts/lw_probe.py C2 64 0.02
set -e
REF=$1; OUT=$(realpath -m $2); shift 2
T=$(mktemp -d)
if [ "$REF" = WORK ]; then mkdir -p $T/tsim_amd $T/include; cp -r tsim_amd/csrc $T/tsim_amd/; cp include/tsim_hip.h $T/include/
else git archive $REF tsim_amd/csrc include | tar -x -C $T; fi
cd $T/tsim_amd/csrc
ls *.hip | xargs -P 8 -I{} hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c {} -o {}.o
for f in *.cpp; do hipcc -x c++ -O3 -std=c++17 -ffp-contract=off -fPIC -msse4.1 -c $f -o $f.o; done
hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $OUT -lrccl
rm -rf $T
echo built $OUT
