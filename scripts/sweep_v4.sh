#!/bin/bash
for gt in 2 4 8; do for blk in 128 256 512; do
  echo -n "GT=$gt BLK=$blk: "
  TSIM_AMD_V4_GT=$gt TSIM_AMD_V4_BLOCK=$blk timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_avg_ms'])"
done; done
