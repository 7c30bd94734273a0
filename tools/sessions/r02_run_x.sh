#!/bin/bash
# what the driver runs at round end (smoke), the multi-rank bench path rehearsed with gloo ranks sharing the GPU, a second fuzz seed
export TMPDIR=/tmp
O=gpurun_out/r02x
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -n 2 $O/smoke.log
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --points 200000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err
echo "gloo2 rc=$?"; tail -c 600 $O/bench_gloo2.json
PYTHONPATH=. timeout -k 10 400 python tools/fuzz.py 300 23 > $O/fuzz23.log 2>&1
echo "fuzz rc=$?"; tail -n 1 $O/fuzz23.log; grep "FAIL" $O/fuzz23.log | cut -c1-500 | head -n 5
