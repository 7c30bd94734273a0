#!/bin/bash
# the round-end rehearsal: build() + smoke() + the default bench line, as the driver runs them
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -1 $O/bench.json
