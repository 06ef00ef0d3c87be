#!/bin/bash
# The other frame sizes through bench.py (measurement aid): the uncropped KITTI frame, 1080p at 256 disparities, 4K at 512.
set -u
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  timeout -k 10 280 python3 bench.py --cpu-frames 0 --no-gather "$@" > gpurun_out/size_$tag.json 2>gpurun_out/size_$tag.err || { echo FAIL $tag; tail -3 gpurun_out/size_$tag.err; exit 1; }
  python3 -c "import json; r=json.load(open('gpurun_out/size_$tag.json')); print('$tag', r['ms_per_step'], 'ms/step', r['value'], r['unit'], 'kernel_ms', r['roofline']['kernel_ms'], 'frac', r['roofline']['frac'])"
}
run kitti_full --size 1242x375 --steps 20 --warmup 3 || exit 1
run hd1080 --size 1920x1072 --max-disp 256 --frames 32 --steps 5 --warmup 1 || exit 1
run uhd4k --size 3840x2160 --max-disp 512 --frames 8 --steps 3 --warmup 1 || exit 1
