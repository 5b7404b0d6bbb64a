#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the default bench command, (2) separate
# --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slots: they do not fit one pass) at the headline
# size and at 4 M envs (working set 1.5 GB >> 256 MB Infinity Cache: the calibration point where
# HBM reads == algorithmic reads).  Run on the GPU box via gpurun; outputs under gpurun_out/<tag>/.
tag=${1:-prof}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2000 --warmup 200 > $O/bench_traced.json 2> $O/bench_traced.err
for E in 65536 4194304; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$E -- python3 $R/bench.py --envs $E --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-rollout > $O/pmc_${c}_$E.json 2> $O/pmc_${c}_$E.err
  done
done
python3 $R/bench.py --steps 2000 --warmup 200 > $O/bench_plain.json 2> $O/bench_plain.err
python3 $R/tools/summarize_profile.py $O
