#!/bin/bash
# usage: tools/ab.sh OUTFILE LINE...   each LINE = "name|ENV=.. ENV=..|bench args"; name "base" = the product library
# One bench.py run per line on ONE box (boxes differ by ~0.1 us per launch); appends "name launch_us frac" to OUTFILE.
out=$1; shift
for line in "$@"; do
  IFS='|' read -r name envs args <<< "$line"
  lib=${name%%:*}
  ( if [ "$lib" != "base" ]; then export ACAS2D_BENCH_LIB=libacas2d_hip_$lib.so; fi
    for kv in $envs; do export "$kv"; done
    python bench.py --no-extra --no-cpu-baseline --no-rollout $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-28s launch_us %.3f  frac %.4f  ms_per_step %.5f  episodes %d' % ('$name $args', d['roofline']['launch_us'], d['roofline']['frac'], d['ms_per_step'], d['config']['episodes_finished']))
" ) | tee -a $out
done
