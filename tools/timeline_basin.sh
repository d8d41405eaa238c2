#!/bin/bash
# timeline of the large-mesh step: rocprofv3 kernel trace of a short run, analysed per step (tools/timeline_analyze.py).  usage: timeline_basin.sh TAG [channel|basin]
set -e
TAG=${1:-r03t}; WL=${2:-basin}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $OUT/trace -o t -- python3 $ROOT/tools/chan_probe.py --workload $WL --levels 3 --steps 12 --warmup 6 > $OUT/trace.log 2>&1
cd $ROOT
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_analyze.py $F k_vel_nodes $OUT/timeline.json > $OUT/timeline.txt
find $OUT -name "*kernel_trace.csv" -delete
tail -70 $OUT/timeline.txt
