#!/bin/bash
# A/B of the left-right configuration: right view from the first sweep's cost values vs a second sweep (measurement aid)
set -u
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --no-gather --variant 9x7lr > gpurun_out/ab_lr_$tag.json 2>gpurun_out/ab_lr.err || { echo FAIL $tag; tail -3 gpurun_out/ab_lr.err; exit 1; }
  python3 -c "import json; r=json.load(open('gpurun_out/ab_lr_$tag.json')); print('$tag', 'ms_step', r['ms_per_step'], 'value', r['value'], 'kernel_ms', r['roofline']['kernel_ms'], r.get('parity'))"
}
run two SVX_LR_TWO_SWEEPS=1
run both2 SVX_SWEEP_OCC_BOTH=2
run both3 SVX_SWEEP_OCC_BOTH=3
