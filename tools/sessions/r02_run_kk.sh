#!/bin/bash
# phase A group size at the reference's default leaf size 10 (and 16): 256 / 512 (default) / 1024 / 2048 points per group
export TMPDIR=/tmp
O=gpurun_out/r02kk
mkdir -p $O
for leaf in 10 16; do
for g in 256 512 1024 2048; do
  HTOOL_PHASE_A_GROUP=$g timeout -k 10 300 python bench.py --leaf $leaf --no-cpu-baseline --no-warm-build > $O/bench_leaf${leaf}_g$g.json 2> $O/bench_leaf${leaf}_g$g.err || exit 1
  echo "leaf $leaf group $g done"
done
done
