set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $OUT/trace -o t -- python3 $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-other --no-large-mesh > $OUT/trace.json 2> $OUT/trace.log
cd $ROOT
python3 tools/timeline.py "$OUT/trace/**/*kernel_trace.csv" 40 > $OUT/timeline_pi.txt
find $OUT -name "*kernel_trace.csv" -delete
tail -5 $OUT/timeline_pi.txt
