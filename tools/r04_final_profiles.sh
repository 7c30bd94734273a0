#!/bin/bash
# final evidence of round 4, part 1 (one gpurun call): bench line, rocprofv3 kernel stats, PMC passes (separate runs), build timelines
# -- 1 M Laplace at leaf 100 (headline) and at the reference's default leaf size 10, and BASELINE C3 (Helmholtz)
export TMPDIR=/tmp
O=gpurun_out/r04final
mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo bench done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err; echo kernel trace done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/fetch.json 2> $O/fetch.err; echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/write.json 2> $O/write.err; echo write done
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/buildprof.log 2>&1; echo buildprof done
# the reference's default leaf size
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_leaf10 -o kt -- python3 bench.py --leaf 10 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_leaf10_under_rocprof.json 2> $O/kt_leaf10.err; echo leaf10 kernel trace done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_leaf10 -o fetch -- python3 bench.py --leaf 10 --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/fetch_leaf10.json 2> $O/fetch_leaf10.err; echo leaf10 fetch done
timeout -k 10 300 python tools/buildprof.py laplace 1000000 3 1e-3 10 > $O/buildprof_leaf10.log 2>&1; echo buildprof leaf10 done
# BASELINE config C3
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o kt -- python3 bench.py --kernel helmholtz --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2> $O/kt_c3.err; echo c3 kernel trace done
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_c3 -o fetch -- python3 bench.py --kernel helmholtz --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/fetch_c3.json 2> $O/fetch_c3.err; echo c3 fetch done
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_c3 -o write -- python3 bench.py --kernel helmholtz --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/write_c3.json 2> $O/write_c3.err; echo c3 write done
timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 > $O/buildprof_c3.log 2>&1; echo buildprof c3 done
# the cluster tree alone (device against host, 1 M points)
timeout -k 10 300 python -m pytest tests/test_gpu_cluster_tree.py -q -m gpu -s -k million > $O/cluster_tree_1m.log 2>&1; echo cluster tree done
find $O -name "*.csv" | head -40
