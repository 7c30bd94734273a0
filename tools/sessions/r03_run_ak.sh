#!/bin/bash
# final evidence of the round: Laplace line + rocprofv3 + PMC passes, then smoke + GPU suite + (in a later call) the rehearsal line
bash tools/r03_final_profiles_laplace.sh || exit 1
O=gpurun_out/r03ak; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.log | tail -2
