#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant kernel for the three bench variants (run on the GPU box); prints one line per variant.
set -u
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 9x7lr 9x7 5x5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_t_${v}_$c
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_t_${v}_$c -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-gather --variant $v > /dev/null 2>&1 || { echo "pass $v $c failed"; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for v in ("9x7lr", "9x7", "5x5"):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = defaultdict(list)
        for f in glob.glob("gpurun_out/pmc_t_%s_%s/**/*counter_collection.csv" % (v, c), recursive=True):
            for row in csv.DictReader(open(f)):
                if "sweep_kernel" in row["Kernel_Name"]:
                    agg[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
        out[c] = {k: round(sum(x) / len(x), 1) for k, x in agg.items()}
    print(v, out)
PY
