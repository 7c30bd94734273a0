#!/bin/bash
# replay of the fuzz case that missed its tolerance (next to the CPU oracle; with the workgroup ACA kernel only as well); build timeline after the host-side changes
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
PYTHONPATH=. timeout -k 10 600 python tools/fuzz.py 330 11 534 > $O/replay534.log 2>&1
echo "replay rc=$?"
tail -n 4 $O/replay534.log | cut -c1-300
HTOOL_ACA_KERNEL=block PYTHONPATH=. timeout -k 10 600 python tools/fuzz.py 330 11 534 > $O/replay534_block.log 2>&1
tail -n 4 $O/replay534_block.log | cut -c1-300
timeout -k 10 200 python tools/buildprof.py laplace 1000000 3 2> $O/bp.log || exit 1
grep -E "native build timing|timeline" $O/bp.log | tail -n 2
