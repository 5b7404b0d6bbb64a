#!/bin/bash
# usage: tools/run_variants.sh OUTFILE [bench args...] -- NAME...   (NAME "" = the product library)
# Per variant library (make -C gym-acas2d_amd/csrc variant NAME=x FLAGS=...) one bench.py run; prints launch_us.
out=$1; shift
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
for name in "$@"; do
  if [ "$name" = "base" ]; then unset ACAS2D_BENCH_LIB; else export ACAS2D_BENCH_LIB=libacas2d_hip_$name.so; fi
  python bench.py --no-extra --no-cpu-baseline --no-rollout "${args[@]}" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-14s launch_us %.3f  frac %.4f  ms_per_step %.5f' % ('$name', d['roofline']['launch_us'], d['roofline']['frac'], d['ms_per_step']))
" | tee -a $out
done
