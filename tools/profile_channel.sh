#!/bin/bash
# One GPU-box call: CORE2-class channel workload -- probe (ms/step, monitor, per-kernel HIP-event times), rocprofv3 kernel stats,
# PMC passes (FETCH_SIZE, WRITE_SIZE, SQ occupancy/stall counters; separate runs, kernel trace only).  usage: profile_channel.sh TAG [LEVELS] [channel|basin]
set -e
TAG=${1:-r03x}; LEV=${2:-3}; WL=${3:-channel}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 tools/chan_probe.py --workload $WL --levels $LEV --steps 60 --warmup 20 --monitor-every 20 --kernels > $OUT/probe.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o s -- python3 $ROOT/tools/chan_probe.py --workload $WL --levels $LEV --steps 30 --warmup 10 > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o f -- python3 $ROOT/tools/chan_probe.py --workload $WL --levels $LEV --steps 6 --warmup 4 > $OUT/pmc_f.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o w -- python3 $ROOT/tools/chan_probe.py --workload $WL --levels $LEV --steps 6 --warmup 4 > $OUT/pmc_w.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc_sq -o q -- python3 $ROOT/tools/chan_probe.py --workload $WL --levels $LEV --steps 6 --warmup 4 > $OUT/pmc_sq.log 2>&1
cd $ROOT
F=$(find $OUT/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $OUT/pmc_summary.json ${WL}_r$LEV$([ $WL = basin ] && echo _default)
Q=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
python3 tools/pmc_sq_summary.py $Q $OUT/pmc_sq_summary.json
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT $OUT/stats
