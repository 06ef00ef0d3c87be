#!/bin/bash
# Every measurement the design notes quote, in one go (run on the GPU box): tools/final_measure.sh <tag>
set -u
TAG=${1:-fin}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
run() { name=$1; shift; timeout -k 10 280 python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo FAIL $name; tail -3 $O/bench_$name.err; exit 1; }
  python3 -c "import json; r=json.load(open('$O/bench_$name.json')); print('$name', r['ms_per_step'], 'ms/step', r['value'], 'kernel_ms', r['roofline']['kernel_ms'], 'frac', r['roofline']['frac'], r.get('parity'))"; }
run 9x7lr || exit 1
run 9x7 --variant 9x7 --cpu-frames 0 || exit 1
run 5x5 --variant 5x5 --cpu-frames 0 || exit 1
run vga_5x5 --variant 5x5 --size 640x480 --max-disp 64 --cpu-frames 0 --no-gather || exit 1
run vector_sweep --matrix-cores 0 --cpu-frames 0 || exit 1
bash tools/bench_sizes.sh || exit 1
cp gpurun_out/size_*.json $O/ 2>/dev/null
timeout -k 10 200 python3 tools/time_entries.py > $O/entry_times.txt 2>&1 || exit 1
cat $O/entry_times.txt
bash tools/prof_mx.sh $TAG > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
head -12 gpurun_out/prof_$TAG/summary.txt
bash tools/traffic_all.sh > $O/traffic.log 2>&1 || { tail -5 $O/traffic.log; exit 1; }
tail -4 $O/traffic.log
