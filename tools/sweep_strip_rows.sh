#!/bin/bash
# Sweeps of the strip length of the matrix-core search (measurement aid, run on the GPU box): time per launch for forced output rows per strip
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for r in 0 61 73 91 122 182; do
    SVX_MX_STRIP_ROWS=$r timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-gather > gpurun_out/sw.json 2>gpurun_out/sw.err || { echo FAIL $r; tail -3 gpurun_out/sw.err; exit 1; }
    python3 -c "import json; r=json.load(open('gpurun_out/sw.json')); print('strip_rows', $r, 'ms_step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms'])"
  done
done
