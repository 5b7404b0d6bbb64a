cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $O/p -- python3 $R/bench.py --steps 40 --warmup 10 --launch eager --no-extra --no-cpu-baseline --no-rollout > $O/out.json 2> $O/err.txt
f=$(find $O/p -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,statistics,re
agg={}
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(r"step_kernel<float, 4, 2, true, true, true(, false)+>", r["Kernel_Name"]):
        agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, len(v), statistics.mean(v[len(v)//4:]))
PY
rm -rf $O/p
