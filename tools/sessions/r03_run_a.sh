#!/bin/bash
# round 3, session a: the library's own exchange with 2 / 3 ranks on one GPU, rank r of 8 through a ctypes hook, the compaction kernel,
# the multi-rank bench rehearsal "inside the library"
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_distributed.py tests/test_gpu_capi_ctypes.py -m gpu -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 15 $O/tests.log
[ $rc -eq 0 ] || exit 1
for w in 2 3; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2961$w bench.py --gpus $w --backend gloo --points 200000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gloo$w.json 2> $O/bench_gloo$w.err
  rc=$?; echo "gloo$w rc=$rc"; tail -c 900 $O/bench_gloo$w.json
  [ $rc -eq 0 ] || { tail -n 20 $O/bench_gloo$w.err; exit 1; }
done
