#!/bin/bash
# numbers quoted in DESIGN.md section 5: noise-level sensitivity and the other shapes (run on the GPU box)
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.3g shots/s' % d['value'], '%.1f us/step' % (d['ms_per_step']*1e3), 'dominant %.1f us' % (r['kernel_avg_ms']*1e3), r['kernel'][:40])"; }
for pb in 0.0 0.001 0.005 0.02 0.05 0.1 0.3; do
  timeout 120 python bench.py --no-cpu-baseline --no-full-leg --p-bit $pb 2>/dev/null | tail -1 | pr "C2 p_bit=$pb"
done
for c in C3 C4 C5; do
  sh=1000000; [ $c = C4 ] && sh=200000
  timeout 200 python bench.py --no-cpu-baseline --config $c --shots $sh 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', '%.3g shots/s' % d['value'], '%.1f us/step' % (d['ms_per_step']*1e3), r['kernel'][:60], 'full-kernel-only:', d.get('full_kernel_only'))"
done
python scripts/info.py C2 C3 C4 C5
