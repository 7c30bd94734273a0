#!/bin/bash
# round 3, session k: XCD-aware order of the phase-B tiles against heaviest-first (A/B by environment variable, same box), product tests
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_hmatrix.py tests/test_gpu_symmetric_storage.py tests/test_gpu_transposed.py -m gpu -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-warm-build > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
for leaf in 10 100 16; do
  for ord in heavy xcd heavy xcd; do
    HTOOL_TILE_ORDER=$ord run bench_leaf${leaf}_${ord}_$RANDOM --leaf $leaf
  done
done
HTOOL_TILE_ORDER=xcd run bench_helm_xcd --kernel helmholtz
HTOOL_TILE_ORDER=heavy run bench_helm_heavy --kernel helmholtz
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch10 -o fetch -- python3 bench.py --leaf 10 --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/fetch10.json 2> $O/fetch10.err
echo "fetch10 rc=$?"
python - <<'PY'
import csv, glob, json
for f in sorted(glob.glob("gpurun_out/r03k/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f.split("/")[-1].ljust(40), "value", round(d["value"]), "ms", round(d["ms_per_step"],3), "phaseB us", round(r["launch_us"],1), "frac", round(r["frac"],4), "phaseA", round(r["phase_a_achieved"]))
f=glob.glob("gpurun_out/r03k/fetch10/**/*counter_collection.csv", recursive=True)
tot={}
for r in csv.DictReader(open(f[0])):
    k=(r["Kernel_Name"][:60], int(r["Grid_Size"]))
    tot.setdefault(k, []).append(float(r["Counter_Value"]))
for k,v in sorted(tot.items(), key=lambda kv: -sum(kv[1]))[:4]:
    if "tile_gemv" in k[0]: print(k, len(v), "FETCH x2 =", round(2*1024*sum(v)/len(v)/1e9,2), "GB")
PY
