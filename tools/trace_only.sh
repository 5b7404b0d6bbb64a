#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default headline run only (the first pass of profile_round.sh), twice, plus the
# untraced run of the same box; keeps the kernel-stats tables and the bench lines under gpurun_out/<tag>/.
tag=${1:-trace}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 2000 --warmup 200 --no-extra --no-cpu-baseline --no-rollout 2>/dev/null | grep "^{" > $O/bench_plain.json
for i in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$i -- python3 $R/bench.py --steps 2000 --warmup 200 --no-extra --no-cpu-baseline --no-rollout 2>/dev/null | grep "^{" > $O/bench_traced$i.json
  f=$(find $O/trace$i -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats$i.csv; rm -rf $O/trace$i
  grep "step_kernel<float, 4, 2" $O/kernel_stats$i.csv | cut -d, -f1-12 | rev | cut -d'"' -f1 | rev
done
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
for n in ("bench_plain", "bench_traced1", "bench_traced2"):
    d = json.loads(open("%s/%s.json" % (o, n)).read()); print(n, "launch_us %.3f" % d["roofline"]["launch_us"], "ms_per_step %.5f" % d["ms_per_step"])
PY
