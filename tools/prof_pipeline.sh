#!/bin/bash
# Kernel trace of one pipeline frame stream (tools/bench_pipeline.py --frames-only): which kernels the device spends a frame in (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${OUT:-r3a}
rm -rf gpurun_out/${OUT:-r3a}/prof_pipe
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${OUT:-r3a}/prof_pipe -- python3 tools/bench_pipeline.py --single 2 > gpurun_out/${OUT:-r3a}/prof_pipe.log 2>&1 || { tail -5 gpurun_out/${OUT:-r3a}/prof_pipe.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/%s/prof_pipe/*/*kernel_stats.csv" % os.environ.get("OUT", "r3a"))[0]
tot = 0
for r in csv.DictReader(open(f)):
    print("%-70s calls %-5s avg_us %8.1f total_us %9.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
    tot += float(r["TotalDurationNs"])
print("total kernel time us", tot / 1e3)
PY
