#!/bin/bash
# ACAS2D_SHAPE sweep for the fused-rollout kernel.  usage: tools/sweep_rollout.sh <envs> <traffic> <out.log> shape...
envs=$1; traffic=$2; out=$3; shift 3
for sh in "$@"; do
  ACAS2D_SHAPE="$sh" timeout -k 10 120 python bench.py --envs "$envs" --traffic "$traffic" --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        f = json.loads(l).get('fused_rollout', {})
        print('shape $sh envs=$envs N=$traffic: rollout %.4g env-steps/s  %.2f us/step  %.0f GB/s' % (f.get('value', 0), f.get('launch_ms', 0) * 1e3 / max(f.get('steps_per_launch', 1), 1), f.get('achieved_GBps', 0)), f.get('error', ''))
" >> "$out"
done
