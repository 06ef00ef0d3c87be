#!/bin/bash
# rocprofv3 kernel-trace stats and counter passes of bench.py, with the counters of the matrix-core search (run on the GPU box).
# Usage: tools/prof_mx.sh <tag> [value of SVX_MX; default: the library's own choice of kernels]
set -u
TAG=${1:-mx}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ -n "${2:-}" ]; then export SVX_MX=$2; fi   # (the default a new context starts from)
BENCH="python3 bench.py --cpu-frames 0"      # the default run: 20 warm-up and 100 timed steps
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $OUT/pmc_$name.log; }
done
python3 tools/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
