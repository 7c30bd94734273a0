#!/bin/bash
# size sweep of the Laplace operator on the final code of round 3 (one box, back to back): cold and second build, product; 2 M points = 185 GB of panels
export TMPDIR=/tmp
O=gpurun_out/r03sweep
mkdir -p $O
for n in 100000 250000 500000 1000000 2000000; do
  timeout -k 10 500 python bench.py --points $n --no-cpu-baseline > $O/bench_n$n.json 2> $O/bench_n$n.err || { echo "n=$n failed"; tail -n 5 $O/bench_n$n.err; exit 1; }
  echo "n=$n done"
done
python - <<'PY'
import json
runs=[]
for n in (100000,250000,500000,1000000,2000000):
    d=json.loads(open(f"gpurun_out/r03sweep/bench_n{n}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    runs.append({"n_points": n, "ms_per_product": d["ms_per_step"], "GBps": d["value"], "algorithmic_GB": d["algorithmic_GB"], "rel_err_sampled_rows": d["rel_err_sampled_rows"],
                 "phase_b_frac_of_8TBps": r["frac"], "build_s": d["build_s"], "build_cold_s": d["build_cold_s"], "cluster_tree_s": d["cluster_tree_s"], "setup_s": d["setup_s"], "hmatrix": d["hmatrix"]})
    print(n, round(d["ms_per_step"],3), round(d["value"]), round(r["frac"],3), "build", round(d["build_s"],3), "cold", round(d["build_cold_s"],2), "err", d["rel_err_sampled_rows"])
json.dump({"note": "python bench.py --points N --no-cpu-baseline, one box, back to back (tools/r03_run_sweep.sh); build_s = second build of the process, build_cold_s = first", "runs": runs}, open("gpurun_out/r03sweep/size_sweep.json","w"), indent=1)
PY
