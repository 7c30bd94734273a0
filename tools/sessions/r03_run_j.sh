#!/bin/bash
# round 3, session j: GPU suite on the new pack path, build timing, and where the small-leaf product stands (kernel stats + FETCH_SIZE at leaf 10)
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc"; tail -n 5 $O/gpu_suite.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/bp_laplace.log 2>&1; grep -E "native build timing" $O/bp_laplace.log | sed -e 's/.*block tree/block tree/' | tail -n 3; grep -E "pack batch 1" $O/bp_laplace.log | tail -n 1; grep -E "timeline" $O/bp_laplace.log | tail -n 1 | cut -c1-700
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt10 -o kt -- python3 bench.py --leaf 10 --steps 20 --warmup 3 --no-cpu-baseline --no-warm-build > $O/bench_leaf10_under_rocprof.json 2> $O/kt10.err
echo "kt10 rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch10 -o fetch -- python3 bench.py --leaf 10 --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/fetch10.json 2> $O/fetch10.err
echo "fetch10 rc=$?"
python - <<'PY'
import csv, glob, json
d=json.loads(open("gpurun_out/r03j/bench_leaf10_under_rocprof.json").read().strip().splitlines()[-1])
print("leaf10: value", round(d["value"]), "ms", round(d["ms_per_step"],3), "frac", d["roofline"]["frac"], "alg bytes B", d["roofline"]["algorithmic_bytes_per_launch"], "phaseA", d["roofline"]["phase_a_achieved"], d["roofline"]["other_kernels_us"])
f=glob.glob("gpurun_out/r03j/kt10/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,1))
f=glob.glob("gpurun_out/r03j/fetch10/**/*counter_collection.csv", recursive=True)
tot={}
for r in csv.DictReader(open(f[0])):
    k=(r["Kernel_Name"][:60], int(r["Grid_Size"]))
    tot.setdefault(k, []).append(float(r["Counter_Value"]))
for k,v in sorted(tot.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(k, len(v), "avg FETCH KiB", round(sum(v)/len(v)), "x2x1024 B =", round(2*1024*sum(v)/len(v)/1e9,2), "GB")
PY
