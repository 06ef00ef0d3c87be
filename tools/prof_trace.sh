#!/bin/bash
# rocprofv3 kernel trace of the default bench run only (no counter passes); prints the per-kernel averages (run on the GPU box).
# Usage: tools/prof_trace.sh <tag> [bench arguments]
set -u
TAG=${1:-t}
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-frames 0 "$@" > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
python3 tools/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
head -12 $OUT/summary.txt
