#!/bin/bash
# Shape sweep for the step kernel: ACAS2D_SHAPE="C,G" | "generic,G" -> bench.py (no CPU baseline).
# usage: tools/sweep_shapes.sh <envs> <traffic> <dtype> <out.log> shape...
envs=$1; traffic=$2; dtype=$3; out=$4; shift 4
for sh in "$@"; do
  echo "== shape $sh envs=$envs N=$traffic $dtype $EXTRA" >> "$out"
  ACAS2D_SHAPE="$sh" timeout -k 10 120 python bench.py --envs "$envs" --traffic "$traffic" --dtype "$dtype" \
      --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline --no-rollout $EXTRA 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   value %.4g env-steps/s  launch %.2f us  achieved %.0f GB/s  frac %.3f' % (d['value'], r['launch_us'], r['achieved'], r['frac']))
    elif 'rror' in l: print('  ', l.strip())
" >> "$out"
done
