#!/bin/bash
# One GPU-box call that produces everything profiles/ holds for a round (run from the repo root via gpurun):
#   bench line (with CPU reference), rocprofv3 --output-format csv --kernel-trace --stats of the same command, two PMC passes (FETCH_SIZE,
#   WRITE_SIZE; separate runs, kernel trace only) and their per-kernel summary.  TAG names the outputs.
set -e
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --physics default --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench_default.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o s -- python3 $ROOT/bench.py --steps 500 --warmup 100 --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.log
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_default -o s -- python3 $ROOT/bench.py --physics default --steps 500 --warmup 100 --no-cpu-baseline > $OUT/stats_default_bench.json 2> $OUT/stats_default.log
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o f -- python3 $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_f.log
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o w -- python3 $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_w.log
cd $ROOT
F=$(find $OUT/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $OUT/pmc_summary.json
find $OUT -name "*kernel_trace.csv" -delete       # (large; the stats tables are what profiles/ keeps)
ls -la $OUT $OUT/stats
