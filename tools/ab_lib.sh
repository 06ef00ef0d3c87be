#!/bin/bash
# A/B of two builds of the library on one box (measurement aid): tools/ab_lib.sh <other.so> [bench options]
set -u
cd $GRAFT_REPO_ROOT
OTHER=$1; shift
for rep in 1 2 3; do
  for which in other this; do
    if [ $which = other ]; then export SVX_LIBRARY=$GRAFT_REPO_ROOT/$OTHER; else unset SVX_LIBRARY; fi
    timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-gather "$@" > gpurun_out/ab_$which.json 2>gpurun_out/ab.err || { echo FAIL $which; tail -3 gpurun_out/ab.err; exit 1; }
    python3 -c "import json; r=json.load(open('gpurun_out/ab_$which.json')); print('$which', 'ms_step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms'])"
  done
done
