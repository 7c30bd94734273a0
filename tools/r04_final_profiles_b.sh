#!/bin/bash
# final evidence of round 4 on the final commit, headline workload only (the other files of profiles/r04_* were taken earlier in the round on the
# same product kernels): bench line, rocprofv3 kernel stats, the two PMC passes (separate runs); then the hierarchical LU's figures
export TMPDIR=/tmp
O=gpurun_out/r04final
mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo bench done; tail -c 600 $O/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err; echo kernel trace done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/fetch.json 2> $O/fetch.err; echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/write.json 2> $O/write.err; echo write done
HTOOL_HLU_PROFILE=1 timeout -k 10 120 python tools/hlu_bench.py 12000 100 1e-3 0 > $O/hlu_12k.json 2> $O/hlu_12k.err; echo hlu 12k done
HTOOL_HLU_PROFILE=1 timeout -k 10 200 python tools/hlu_bench.py 500000 100 1e-3 1 8e-3 > $O/hlu_c5_block.json 2> $O/hlu_c5_block.err; echo hlu c5 block done
timeout -k 10 200 python tools/hlu_bench.py 500000 100 1e-3 1 8e-3 > $O/hlu_c5_block_no_profile.json 2> /dev/null; echo hlu c5 block again done
find $O -name "*.csv" | head
