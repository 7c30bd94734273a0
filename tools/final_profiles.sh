#!/bin/bash
# final evidence of the round, part 1: bench line, rocprofv3 kernel stats, PMC passes (separate runs), same command line
set -e
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
echo bench done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
echo kernel trace done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err
echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.json 2> $O/write.err
echo write done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt16 -o kt -- python3 bench.py --rhs 16 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rhs16_under_rocprof.json 2> $O/kt16.err
echo rhs16 trace done
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/buildprof.log 2>&1
echo buildprof done
find $O -name "*.csv" | head -20
