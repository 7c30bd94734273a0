"""Copy the evidence of tools/final_profiles.sh + tools/final_benches.sh from gpurun_out/ (scratch) into profiles/ (tracked),
under the round's names, and derive the PMC traffic file bench.py reads.   python tools/collect_profiles.py"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, F2, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "gpurun_out", "final2"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


def first(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[0] if g else ""


cp(os.path.join(F, "bench.json"), "r02_bench_1m_laplace.json")
cp(os.path.join(F, "bench_under_rocprof.json"), "r02_bench_1m_laplace_under_rocprof.json")
cp(first(os.path.join(F, "kt", "**", "*kernel_stats.csv")), "r02_bench_1m_laplace_kernel_stats.csv")
cp(first(os.path.join(F, "fetch", "**", "*counter_collection.csv")), "r02_pmc_fetch_size_counter_collection.csv")
cp(first(os.path.join(F, "write", "**", "*counter_collection.csv")), "r02_pmc_write_size_counter_collection.csv")
cp(first(os.path.join(F, "kt16", "**", "*kernel_stats.csv")), "r02_bench_1m_laplace_rhs16_kernel_stats.csv")
cp(os.path.join(F, "bench_rhs16_under_rocprof.json"), "r02_bench_1m_laplace_rhs16_under_rocprof.json")
fetch, write = os.path.join(P, "r02_pmc_fetch_size_counter_collection.csv"), os.path.join(P, "r02_pmc_write_size_counter_collection.csv")
if os.path.exists(fetch) and os.path.exists(write):
    out = subprocess.run([sys.executable, os.path.join(P, "derive_pmc_traffic.py"), fetch, write], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(P, "r02_pmc_hbm_traffic_1m_laplace.json"), "w") as f:
        f.write(out)
    print("derived r02_pmc_hbm_traffic_1m_laplace.json")
if os.path.exists(fetch) and os.path.exists(write):
    # the build kernels of the same two passes (one 1 M-point build per bench run, plus the 20 000-point warm-up build): bytes the
    # ACA kernels fetched (8-byte loads per lane: FETCH_SIZE as reported, not doubled) and wrote, against the factors they produced
    import csv
    from collections import defaultdict

    tot = defaultdict(lambda: defaultdict(float))
    for path in (fetch, write):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].replace("void ", "").split("(")[0]
                if name.startswith("hm::aca_") or name.startswith("hm::pack_") or name.startswith("hm::compact_"):
                    tot[name][row["Counter_Name"]] += float(row["Counter_Value"]) * 1024.0
    bench_line = json.loads(open(os.path.join(P, "r02_bench_1m_laplace.json")).read().strip().splitlines()[-1]) if os.path.exists(os.path.join(P, "r02_bench_1m_laplace.json")) else {}
    aca_fetch = sum(v["FETCH_SIZE"] for k, v in tot.items() if k.startswith("hm::aca_"))
    aca_write = sum(v["WRITE_SIZE"] for k, v in tot.items() if k.startswith("hm::aca_"))
    build = {"note": "sums over ALL launches of the build kernels in the two PMC passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline` (one 1 M-point build "
                     "+ the 20 000-point warm-up build each); FETCH_SIZE / WRITE_SIZE in bytes as reported (x 1024), FETCH not doubled (8-byte loads per lane)",
             "per_kernel_bytes": {k: dict(v) for k, v in sorted(tot.items())},
             "aca_kernels_fetch_GB": aca_fetch / 1e9, "aca_kernels_write_GB": aca_write / 1e9}
    with open(os.path.join(P, "r02_pmc_build_1m_laplace.json"), "w") as f:
        json.dump(build, f, indent=1)
        f.write("\n")
    print("derived r02_pmc_build_1m_laplace.json: ACA fetch %.1f GB, write %.1f GB" % (aca_fetch / 1e9, aca_write / 1e9))
if os.path.exists(os.path.join(F, "buildprof.log")):
    with open(os.path.join(F, "buildprof.log")) as f, open(os.path.join(P, "r02_build_timeline_1m_laplace.txt"), "w") as g:
        g.write("# python tools/buildprof.py laplace 1000000 4   (DEBUG log of the native build: stage@seconds marks, ACA rounds, pack batches)\n")
        for line in f:
            if any(k in line for k in ("timeline", "native build timing", "ACA ", "pack batch", "rep ")):
                g.write(line)
    print("wrote r02_build_timeline_1m_laplace.txt")
names = {"bench_rhs1": "r02_bench_1m_laplace_rhs1.json", "bench_rhs8": "r02_bench_1m_laplace_rhs8.json", "bench_rhs16": "r02_bench_1m_laplace_rhs16.json",
         "bench_leaf16": "r02_bench_1m_laplace_leaf16.json", "bench_leaf10": "r02_bench_1m_laplace_leaf10.json", "bench_c2_100k": "r02_bench_c2_100k_laplace_eps1e-4.json",
         "bench_c5_gmres50": "r02_bench_c5_500k_gmres50_1gpu.json", "bench_c3_helmholtz": "r02_bench_c3_1m_helmholtz_c128.json",
         "bench_force_dist": "r02_bench_1m_laplace_library_rccl_one_rank.json", "bench_125k_eager": "r02_bench_125k_graph_replay.json",
         "bench_125k_nograph": "r02_bench_125k_eager.json", "bench_sym_one_triangle": "r02_bench_1m_laplace_sym_one_triangle.json",
         "bench_leaf16_recompressed": "r02_bench_1m_laplace_leaf16_recompressed.json", "per_rank": "r02_per_rank_split_1m_laplace.json",
         "bench_trans_T": "r02_bench_1m_laplace_transposed.json", "bench_helm_rhs16": "r02_bench_c3_1m_helmholtz_rhs16.json",
         "bench_helm_rhs8": "r02_bench_c3_1m_helmholtz_rhs8.json", "bench_helm_trans_C": "r02_bench_c3_1m_helmholtz_conj_transposed.json"}
for src, dst in names.items():
    cp(os.path.join(F2, src + ".json"), dst)
