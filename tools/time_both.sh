cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu_dense64.py -x -q -m gpu --timeout 120 --timeout-method thread -k "both or fused" 2>&1 | tail -3 || exit 1
timeout -k 10 120 python3 tools/time_both.py || exit 1
bash tools/ab_lr.sh
