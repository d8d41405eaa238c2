#!/bin/bash
# One GPU-box call that produces everything profiles/ holds for a round (run from the repo root via gpurun):
#   driver-style bench line (20 steps after 5, with the CPU reference), the long bench line, rocprofv3 --kernel-trace --stats of the
#   same command, two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) and their per-kernel summary.  TAG names the outputs.
set -e
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench.err
python3 bench.py --no-other --no-large-mesh > $OUT/bench.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o s -- python3 $ROOT/bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-other --no-large-mesh > $OUT/stats_bench.json 2> $OUT/stats.log
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o f -- python3 $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-other --no-large-mesh > /dev/null 2> $OUT/pmc_f.log
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o w -- python3 $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-other --no-large-mesh > /dev/null 2> $OUT/pmc_w.log
cd $ROOT
F=$(find $OUT/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $OUT/pmc_summary.json pi_default
find $OUT -name "*kernel_trace.csv" -delete       # (large; the stats tables are what profiles/ keeps)
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT $OUT/stats
