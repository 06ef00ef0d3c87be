#!/bin/bash
# Kernel trace of tools/time_mesh.py: per-kernel times of the device triangulation for each of its point sets (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${OUT:-r3a}
rm -rf gpurun_out/${OUT:-r3a}/prof_mesh
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${OUT:-r3a}/prof_mesh -- python3 tools/time_mesh.py > gpurun_out/${OUT:-r3a}/prof_mesh.log 2>&1 || { tail -5 gpurun_out/${OUT:-r3a}/prof_mesh.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
rows = list(csv.DictReader(open(glob.glob("gpurun_out/%s/prof_mesh/*/*kernel_trace.csv" % os.environ.get("OUT", "r3a"))[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:26] for r in rows]
calls = [k for k, n in enumerate(names) if n.startswith("cell_count")]
for c in calls[1::21]:          # the second call of each of the point sets
    print(" | ".join("%s %.0f" % (names[j], (int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3) for j in range(c, min(c + 12, len(rows))) if not names[j].startswith("__amd")))
PY
