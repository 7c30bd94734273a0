#!/bin/bash
# final evidence of round 4, part 2: the other BASELINE configurations and variants (one JSON line each)
export TMPDIR=/tmp
O=gpurun_out/r04final2
mkdir -p $O
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; echo "$name done"; }
run bench_rhs8 --rhs 8
run bench_rhs16 --rhs 16
run bench_leaf16 --leaf 16
run bench_leaf10 --leaf 10
run bench_c2_100k --points 100000 --eps 1e-4
run bench_c5_gmres50 --points 500000 --gmres 50
run bench_gmres50_62500_one_rank_rccl --points 62500 --gmres 50 --force-dist
run bench_force_dist --force-dist
run bench_125k --points 125000 --steps 300 --warmup 20 --no-phase-timing
run bench_sym_one_triangle --symmetric one-triangle
run bench_sym_rhs16 --symmetric one-triangle --rhs 16
run bench_trans_T --trans T
run bench_trans_T_rhs16 --trans T --rhs 16
run bench_c3_helmholtz --kernel helmholtz --kappa 10
run bench_helm_rhs16 --kernel helmholtz --kappa 10 --rhs 16
run bench_2m --points 2000000
for w in 2 3; do
  HTOOL_BENCH_THREADS=2 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2962$w bench.py --gpus $w --backend gloo --points 200000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gloo$w.json 2> $O/bench_gloo$w.err
  echo "gloo$w done"
done
HTOOL_BENCH_THREADS=2 timeout -k 10 300 python bench.py --points 200000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_200k_one_rank_2threads.json 2> $O/bench_200k_one_rank_2threads.err; echo 200k done
timeout -k 10 300 python tools/dense_lu_62k.py 500000 > $O/dense_lu_62k.log 2>&1; echo dense lu done
echo all done
