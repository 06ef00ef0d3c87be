#!/bin/bash
# Times every variants/libsvx_*.so (tools/build_mx_variant.sh) and the in-tree library on one box, two rounds (measurement aid).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  unset SVX_LIBRARY
  timeout -k 10 120 python3 tools/time_mx.py || exit 1
  for so in variants/libsvx_*.so; do
    SVX_LIBRARY=$GRAFT_REPO_ROOT/$so timeout -k 10 120 python3 tools/time_mx.py || exit 1
  done
done
