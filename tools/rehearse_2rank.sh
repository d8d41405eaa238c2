#!/bin/bash
# One-GPU rehearsal of `bench.py --gpus 2` (correctness and counts only: the two ranks share the box's GPU, gloo carries the messages host-staged).
# usage: rehearse_2rank.sh [TAG] [extra bench args]
TAG=${1:-r03_rehearsal}; shift
FESOM_BENCH_BACKEND=gloo timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 60 --warmup 20 "$@" > gpurun_out/${TAG}_bench_2rank.json 2> gpurun_out/${TAG}_bench_2rank.err
rc=$?
tail -c 3000 gpurun_out/${TAG}_bench_2rank.json; tail -5 gpurun_out/${TAG}_bench_2rank.err
exit $rc
