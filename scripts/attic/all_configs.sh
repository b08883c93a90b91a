#!/bin/bash
# Per-config packer statistics + HIP-event kernel time (no CPU baseline leg).
mkdir -p gpurun_out
python scripts/info.py C1 C2 C3 C4 C5 > gpurun_out/info_all.txt 2>&1
for c in C2 C3 C4 C5; do
  sh=1000000; [ $c = C4 ] && sh=200000
  timeout 300 python bench.py --config $c --shots $sh --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_$c.json
done
