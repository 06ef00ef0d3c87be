#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant kernel for the three bench variants (run on the GPU box); writes profiles-style
# JSON files gpurun_out/traffic_w{1,2,2_both}_mx.json (one PMC pass per counter and variant, nothing else traced).
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
import csv, glob, json
from collections import defaultdict
names = {"9x7lr": ("traffic_w2_both_mx.json", 2, "dense_wta_mx_kernel<5, 2, true>"), "9x7": ("traffic_w2_mx.json", 2, "dense_wta_mx_kernel<5, 2, false>"),
         "5x5": ("traffic_w1_mx.json", 1, "dense_wta_mx_kernel<5, 1, false>")}
for v, (fname, words, kernel) in names.items():
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("gpurun_out/pmc_t_%s_%s/**/*counter_collection.csv" % (v, c), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Kernel_Name"].replace("void ", "").startswith(kernel):
                    vals.append(float(row["Counter_Value"]))
        out[c] = round(sum(vals) / max(len(vals), 1), 1)
        out[c + "_n"] = len(vals)
    rec = {"kernel": "dense_wta_mx_kernel", "template": kernel, "words": words, "variant": v,
           "source": "tools/traffic_all.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) on `python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --variant %s`" % v,
           "fetch_size_kb": out["FETCH_SIZE"], "write_size_kb": out["WRITE_SIZE"], "dispatches_averaged": out["FETCH_SIZE_n"],
           "fetch_correction": 2.0,
           "note": "KB per launch as counted.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B; calibrated on this library's own "
                   "4-B-per-lane streams: mx_lr_check_kernel reads the partial keys and the left map once each (581 + 116 MB at the bench "
                   "size) and reports 354 MB, the search kernel reads 464 MB of census words and reports 240 MB.  Bytes = 2 x fetch + write; "
                   "WRITE_SIZE is exact (census 464 MB written, 453 MB counted without the zero border)"}
    json.dump(rec, open("gpurun_out/" + fname, "w"), indent=1)
    print(v, out)
PY
