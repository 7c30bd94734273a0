#!/bin/bash
O=gpurun_out/h9
mkdir -p $O
for arg in none recompress; do
OMP_NUM_THREADS=4 timeout -k 10 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29733 tools/hlu_example_probe.py $arg > $O/probe_$arg.log 2>&1
echo "== $arg rc=$?"; grep "^rank\|Error\|error" $O/probe_$arg.log | head -12 | cut -c1-300
done
