#!/bin/bash
# final evidence of the round, part 2: the other BASELINE configurations and variants (one JSON line each)
set -e
export TMPDIR=/tmp
O=gpurun_out/final2
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; echo "$name done"; }
run bench_rhs1
run bench_rhs8 --rhs 8
run bench_rhs16 --rhs 16
run bench_leaf16 --leaf 16
run bench_leaf10 --leaf 10
run bench_c2_100k --points 100000 --eps 1e-4
run bench_c5_gmres50 --points 500000 --gmres 50
run bench_force_dist --force-dist
run bench_125k_eager --points 125000 --steps 300 --warmup 20 --no-phase-timing
HTOOL_PRODUCT_GRAPH=0 timeout -k 10 300 python bench.py --points 125000 --steps 300 --warmup 20 --no-phase-timing --no-cpu-baseline > $O/bench_125k_nograph.json 2> $O/bench_125k_nograph.err
run bench_sym_one_triangle --symmetric one-triangle
run bench_leaf16_recompressed --leaf 16 --recompress
run bench_trans_T --trans T
timeout -k 10 300 python tools/per_rank.py > $O/per_rank.json 2> $O/per_rank.err || true
echo all done
