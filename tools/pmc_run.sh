#!/bin/bash
# rocprofv3 PMC pass(es) over a short eager bench run; counters in their own runs (no tracing).
# usage: tools/pmc_run.sh <tag> "<counters>" [bench args...]
tag=$1; counters=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $counters --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 20 --launch eager --no-cpu-baseline --no-rollout "$@" > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if 'step_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    v = v[len(v)//3:]
    print('%-28s n=%d mean=%.4g' % (k, len(v), sum(v)/len(v)))
PY
