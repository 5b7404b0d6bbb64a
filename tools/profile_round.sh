#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the default bench command, (2) separate
# --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slots: they do not fit one pass) at the headline
# size and at 4 M envs (working set 1.5 GB >> 256 MB Infinity Cache: the calibration point where
# HBM reads == algorithmic reads).  Run on the GPU box via gpurun; outputs under gpurun_out/<tag>/.
tag=${1:-prof}
export ACAS2D_BUILD_LABEL=${2:-unlabelled}      # goes into traffic.json: which build the counter passes were taken on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2000 --warmup 200 --no-extra --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err
for E in 65536 4194304; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$E -- python3 $R/bench.py --envs $E --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-rollout --no-extra > $O/pmc_${c}_$E.json 2> $O/pmc_${c}_$E.err
  done
done
# the float64 builds (parity mode EXACT, and FAST): kernel-trace statistics + instruction-mix counters
for m in exact fast; do
  fm=""; [ $m = fast ] && fm="--fast-math"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f64_$m -- python3 $R/bench.py --dtype f64 $fm --steps 1000 --warmup 100 --no-extra --no-rollout --no-cpu-baseline > $O/bench_f64_$m.json 2> $O/bench_f64_$m.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq_f64_$m -- python3 $R/bench.py --dtype f64 $fm --steps 40 --warmup 10 --launch eager --no-extra --no-cpu-baseline --no-rollout > $O/pmc_sq_f64_$m.json 2> $O/pmc_sq_f64_$m.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq_f32 -- python3 $R/bench.py --steps 40 --warmup 10 --launch eager --no-extra --no-cpu-baseline --no-rollout > $O/pmc_sq_f32.json 2> $O/pmc_sq_f32.err
python3 $R/bench.py --steps 2000 --warmup 200 > $O/bench_plain.json 2> $O/bench_plain.err
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python3 $R/tools/summarize_profile.py $O
# keep what is judged (gpurun merges at most 64 MiB back): the summary, the bench lines, the kernel-stats tables and
# the step kernel's per-dispatch counter values; drop the raw trace / counter directories
mkdir -p $O/keep
cp $O/summary.json $O/traffic.json $O/bench_*.json $O/keep/ 2>/dev/null
for t in trace trace_f64_exact trace_f64_fast; do
  f=$(find $O/$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/keep/${t}_kernel_stats.csv
done
for dd in $O/pmc_*/; do
  n=$(basename $dd); f=$(find $dd -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && (head -1 "$f"; grep step_kernel "$f" | tail -400) > $O/keep/${n}_step_kernel.csv
done
rm -rf $O/trace $O/trace_f64_* $O/pmc_*
