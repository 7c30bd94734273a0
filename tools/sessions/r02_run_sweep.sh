#!/bin/bash
# size sweep of the Laplace operator (one box, back to back) and BASELINE config C1 on the GPU (launch-bound: eager vs hipGraph replay)
export TMPDIR=/tmp
O=gpurun_out/sweep
mkdir -p $O
for n in 100000 250000 500000 1000000 2000000; do
  timeout -k 10 400 python bench.py --points $n --no-cpu-baseline --no-warm-build > $O/bench_n$n.json 2> $O/bench_n$n.err || exit 1
  echo "n=$n done"
done
timeout -k 10 300 python bench.py --points 10000 --leaf 50 --symmetric one-triangle --kernel inv_delta --no-cpu-baseline --no-warm-build --steps 500 --warmup 50 > $O/bench_c1.json 2> $O/bench_c1.err || exit 1
timeout -k 10 300 python bench.py --points 10000 --leaf 50 --symmetric one-triangle --kernel inv_delta --no-cpu-baseline --no-warm-build --steps 500 --warmup 50 --no-phase-timing > $O/bench_c1_graph.json 2> $O/bench_c1_graph.err || exit 1
HTOOL_PRODUCT_GRAPH=0 timeout -k 10 300 python bench.py --points 10000 --leaf 50 --symmetric one-triangle --kernel inv_delta --no-cpu-baseline --no-warm-build --steps 500 --warmup 50 --no-phase-timing > $O/bench_c1_eager.json 2> $O/bench_c1_eager.err || exit 1
echo done
