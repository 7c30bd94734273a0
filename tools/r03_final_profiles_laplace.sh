#!/bin/bash
# the Laplace half of tools/r03_final_profiles.sh again, on the final code of the round (the fingerprinted sources changed after the first pass)
set -e
export TMPDIR=/tmp
O=gpurun_out/r03final
mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
echo bench done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
echo kernel trace done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/fetch.json 2> $O/fetch.err
echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/write.json 2> $O/write.err
echo write done
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/buildprof.log 2>&1
echo buildprof done
timeout -k 10 500 python bench.py > $O/bench_after_pmc.json 2> $O/bench_after_pmc.err
echo second bench done
