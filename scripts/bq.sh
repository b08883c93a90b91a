#!/bin/bash
# usage: scripts/bq.sh <bench args...>  - bench.py without the context legs, the figures that matter on one line
python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/tmp/bq.err | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
if not t: print(open('/tmp/bq.err').read()[-2000:]); sys.exit(1)
d=json.loads(t[-1]); r=d['roofline']
print('steps',d['steps'],'value %.3e'%d['value'],'ms/step %.5f'%d['ms_per_step'],'kernel_ms %.4f'%r['kernel_avg_ms'],'batches/launch',r.get('batches_per_launch'),'enqueue_ms/step %.5f'%r['host_enqueue_ms_per_step'],'reps',[round(x,5) for x in d['repeat_ms_per_step']['all']])"
