#!/bin/bash
# Passes of nine tiles (SVX_MX_WIDE_TILES=1) against passes of five (0) beyond 256 disparities, on one box (measurement aid)
cd $GRAFT_REPO_ROOT
for cfg in "3840x2160 512 8" "1920x1072 512 16" "3840x2160 480 8" "3840x2160 384 8" "3840x2160 800 4" "3840x2160 1024 4"; do
  set -- $cfg
  for rep in 1 2; do
    for w in 0 1; do
      SVX_MX_WIDE_TILES=$w timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-gather --size $1 --max-disp $2 --frames $3 > gpurun_out/wide.json 2>gpurun_out/wide.err || { tail -3 gpurun_out/wide.err; exit 1; }
      python3 -c "import json; r=json.loads(open('gpurun_out/wide.json').read().strip().split('\n')[-1]); print('$1 D=$2 frames=$3 wide', $w, 'ms_step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms'])"
    done
  done
done
