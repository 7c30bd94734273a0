"""Copy the evidence of tools/r04_final_profiles.sh + tools/r04_final_benches.sh (round 3: r03_*; `python tools/collect_profiles.py r03`) from gpurun_out/ (scratch) into profiles/ (tracked),
under the round's names, and derive the PMC traffic file bench.py reads.   python tools/collect_profiles.py
(round 2: tools/final_profiles.sh + tools/final_benches*.sh, prefix r02, directories gpurun_out/final and final2)"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
F, F2, P = os.path.join(ROOT, "gpurun_out", R + "final"), os.path.join(ROOT, "gpurun_out", R + "final2"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


def first(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[0] if g else ""


cp(os.path.join(F, "bench.json"), R + "_bench_1m_laplace.json")
cp(os.path.join(F, "bench_under_rocprof.json"), R + "_bench_1m_laplace_under_rocprof.json")
cp(first(os.path.join(F, "kt", "**", "*kernel_stats.csv")), R + "_bench_1m_laplace_kernel_stats.csv")
cp(first(os.path.join(F, "fetch", "**", "*counter_collection.csv")), R + "_pmc_fetch_size_counter_collection.csv")
cp(first(os.path.join(F, "write", "**", "*counter_collection.csv")), R + "_pmc_write_size_counter_collection.csv")
cp(first(os.path.join(F, "kt16", "**", "*kernel_stats.csv")), R + "_bench_1m_laplace_rhs16_kernel_stats.csv")
cp(os.path.join(F, "bench_rhs16_under_rocprof.json"), R + "_bench_1m_laplace_rhs16_under_rocprof.json")
fetch, write = os.path.join(P, R + "_pmc_fetch_size_counter_collection.csv"), os.path.join(P, R + "_pmc_write_size_counter_collection.csv")
if os.path.exists(fetch) and os.path.exists(write):
    out = subprocess.run([sys.executable, os.path.join(P, "derive_pmc_traffic.py"), fetch, write], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(P, R + "_pmc_hbm_traffic_1m_laplace.json"), "w") as f:
        f.write(out)
    print("derived", R + "_pmc_hbm_traffic_1m_laplace.json")
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
    bench_line = json.loads(open(os.path.join(P, R + "_bench_1m_laplace.json")).read().strip().splitlines()[-1]) if os.path.exists(os.path.join(P, R + "_bench_1m_laplace.json")) else {}
    aca_fetch = sum(v["FETCH_SIZE"] for k, v in tot.items() if k.startswith("hm::aca_"))
    aca_write = sum(v["WRITE_SIZE"] for k, v in tot.items() if k.startswith("hm::aca_"))
    build = {"note": "sums over ALL launches of the build kernels in the two PMC passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build` (ONE 1 M-point build "
                     "each); FETCH_SIZE / WRITE_SIZE in bytes as reported (x 1024), FETCH not doubled (8-byte loads per lane)",
             "per_kernel_bytes": {k: dict(v) for k, v in sorted(tot.items())},
             "aca_kernels_fetch_GB": aca_fetch / 1e9, "aca_kernels_write_GB": aca_write / 1e9}
    with open(os.path.join(P, R + "_pmc_build_1m_laplace.json"), "w") as f:
        json.dump(build, f, indent=1)
        f.write("\n")
    print("derived " + R + "_pmc_build_1m_laplace.json: ACA fetch %.1f GB, write %.1f GB" % (aca_fetch / 1e9, aca_write / 1e9))
if os.path.exists(os.path.join(F, "buildprof.log")):
    with open(os.path.join(F, "buildprof.log")) as f, open(os.path.join(P, R + "_build_timeline_1m_laplace.txt"), "w") as g:
        g.write("# python tools/buildprof.py laplace 1000000 4   (DEBUG log of the native build: stage@seconds marks, ACA rounds, pack batches)\n")
        for line in f:
            if any(k in line for k in ("timeline", "native build", "ACA ", "pack batch", "device-resident", "rep ")):
                g.write(line)
    print("wrote", R + "_build_timeline_1m_laplace.txt")
names = {"bench_rhs1": R + "_bench_1m_laplace_rhs1.json", "bench_rhs8": R + "_bench_1m_laplace_rhs8.json", "bench_rhs16": R + "_bench_1m_laplace_rhs16.json",
         "bench_leaf16": R + "_bench_1m_laplace_leaf16.json", "bench_leaf10": R + "_bench_1m_laplace_leaf10.json", "bench_c2_100k": R + "_bench_c2_100k_laplace_eps1e-4.json",
         "bench_c5_gmres50": R + "_bench_c5_500k_gmres50_1gpu.json", "bench_c3_helmholtz": R + "_bench_c3_1m_helmholtz_c128.json",
         "bench_force_dist": R + "_bench_1m_laplace_library_rccl_one_rank.json", "bench_125k_eager": R + "_bench_125k_graph_replay.json",
         "bench_125k_nograph": R + "_bench_125k_eager.json", "bench_sym_one_triangle": R + "_bench_1m_laplace_sym_one_triangle.json",
         "bench_leaf16_recompressed": R + "_bench_1m_laplace_leaf16_recompressed.json", "per_rank": R + "_per_rank_split_1m_laplace.json",
         "bench_trans_T": R + "_bench_1m_laplace_transposed.json", "bench_helm_rhs16": R + "_bench_c3_1m_helmholtz_rhs16.json",
         "bench_helm_rhs8": R + "_bench_c3_1m_helmholtz_rhs8.json", "bench_helm_trans_C": R + "_bench_c3_1m_helmholtz_conj_transposed.json"}
names.update({"bench_125k": R + "_bench_125k_graph_replay.json", "bench_gmres50_62500_one_rank_rccl": R + "_gmres50_62500_one_rank_rccl.json",
              "bench_gmres50_500k_one_rank_rccl": R + "_gmres50_500k_one_rank_rccl.json", "bench_gloo2": R + "_rehearsal_gloo2_200k_inside_library.json",
              "bench_gloo3": R + "_rehearsal_gloo3_200k_inside_library.json"})
names.update({"bench_sym_rhs16": R + "_bench_1m_laplace_sym_one_triangle_rhs16.json", "bench_trans_T_rhs16": R + "_bench_1m_laplace_transposed_rhs16.json",
              "bench_helm_sym_rhs16": R + "_bench_c3_1m_helmholtz_sym_one_triangle_rhs16.json"})
for src, dst in names.items():
    cp(os.path.join(F2, src + ".json"), dst)
# the 16-wide one-triangle sweep under the profiler, the matrix-core rate its fused phase B is priced against, the fuzz run of the final code
cp(first(os.path.join(F, "kt_sym16", "**", "*kernel_stats.csv")), R + "_bench_1m_laplace_sym_rhs16_kernel_stats.csv")
cp(os.path.join(F, "bench_sym_rhs16_under_rocprof.json"), R + "_bench_1m_laplace_sym_rhs16_under_rocprof.json")
cp(os.path.join(F, "mfma_f64_rate.txt"), R + "_mfma_f64_rate.txt")
cp(os.path.join(F2, "dense_lu_62k.log"), R + "_dense_device_lu_62500.txt")
# BASELINE config C3 under the profiler (kernel stats + the two PMC passes), and its build timeline
cp(os.path.join(F, "bench_c3_under_rocprof.json"), R + "_bench_c3_1m_helmholtz_under_rocprof.json")
cp(first(os.path.join(F, "kt_c3", "**", "*kernel_stats.csv")), R + "_bench_c3_1m_helmholtz_kernel_stats.csv")
cp(first(os.path.join(F, "fetch_c3", "**", "*counter_collection.csv")), R + "_pmc_c3_fetch_size_counter_collection.csv")
cp(first(os.path.join(F, "write_c3", "**", "*counter_collection.csv")), R + "_pmc_c3_write_size_counter_collection.csv")
fc3, wc3 = os.path.join(P, R + "_pmc_c3_fetch_size_counter_collection.csv"), os.path.join(P, R + "_pmc_c3_write_size_counter_collection.csv")
if os.path.exists(fc3) and os.path.exists(wc3):
    import csv
    from collections import defaultdict

    tot = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in (fc3, wc3):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].replace("void ", "").split("(")[0]
                e = tot[name][row["Counter_Name"]]
                e[0] += float(row["Counter_Value"]) * 1024.0
                e[1] += 1
    out = {"note": "BASELINE config C3 (1 M-point Helmholtz, complex128): sums over all launches of every kernel in the two PMC passes of `bench.py --kernel helmholtz --steps 3 "
                   "--warmup 1 --no-cpu-baseline --no-warm-build` (ONE build, 4 products); FETCH_SIZE / WRITE_SIZE x 1024 as reported; the product kernels stream 16 B per lane: "
                   "their read bytes are 2 x FETCH_SIZE (MI355X_MICROARCH.md), given as hbm_read_GB_per_launch",
           "per_kernel": {}}
    for k, v in sorted(tot.items()):
        e = {c: {"bytes": b, "launches": n} for c, (b, n) in v.items()}
        if k.startswith("hm::tile_gem") and "FETCH_SIZE" in v:
            e["hbm_read_GB_per_launch"] = 2.0 * v["FETCH_SIZE"][0] / max(v["FETCH_SIZE"][1], 1) / 1e9
        out["per_kernel"][k] = e
    out["aca_kernels_fetch_GB"] = sum(v["FETCH_SIZE"][0] for k, v in tot.items() if k.startswith("hm::aca_") and "FETCH_SIZE" in v) / 1e9
    out["aca_kernels_write_GB"] = sum(v["WRITE_SIZE"][0] for k, v in tot.items() if k.startswith("hm::aca_") and "WRITE_SIZE" in v) / 1e9
    with open(os.path.join(P, R + "_pmc_c3_1m_helmholtz.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("derived", R + "_pmc_c3_1m_helmholtz.json")
if os.path.exists(os.path.join(F, "buildprof_c3.log")):
    with open(os.path.join(F, "buildprof_c3.log")) as f, open(os.path.join(P, R + "_build_timeline_c3_1m_helmholtz.txt"), "w") as g:
        g.write("# python tools/buildprof.py helmholtz 1000000 2   (DEBUG log of the native build: stage@seconds marks, ACA rounds, pack batches; buildprof waits only 3 s between builds, "
                "so the second build's allocations wait for the driver's wipe of the first one's 207 GB: see `pack` -- the bench line's build_s is the figure without that)\n")
        for line in f:
            if any(k in line for k in ("timeline", "native build", "ACA ", "pack batch", "device-resident", "rep ")):
                g.write(line)
    print("wrote", R + "_build_timeline_c3_1m_helmholtz.txt")

# ---- round 4 additions: the reference's default leaf size under the profiler, the cluster tree on the GPU, the 16-wide counter passes
cp(os.path.join(F, "bench_leaf10_under_rocprof.json"), R + "_bench_1m_laplace_leaf10_under_rocprof.json")
cp(first(os.path.join(F, "kt_leaf10", "**", "*kernel_stats.csv")), R + "_bench_1m_laplace_leaf10_kernel_stats.csv")
cp(first(os.path.join(F, "fetch_leaf10", "**", "*counter_collection.csv")), R + "_pmc_leaf10_fetch_size_counter_collection.csv")
if os.path.exists(os.path.join(F, "buildprof_leaf10.log")):
    with open(os.path.join(F, "buildprof_leaf10.log")) as f, open(os.path.join(P, R + "_build_timeline_1m_laplace_leaf10.txt"), "w") as g:
        g.write("# python tools/buildprof.py laplace 1000000 3 1e-3 10   (the reference's default maximal_leaf_size; DEBUG log of the native build)\n")
        for line in f:
            if any(k in line for k in ("timeline", "native build", "ACA ", "pack batch", "device-resident", "rep ")):
                g.write(line)
    print("wrote", R + "_build_timeline_1m_laplace_leaf10.txt")
cp(os.path.join(F, "cluster_tree_1m.log"), R + "_cluster_tree_1m_device_vs_host.txt")
for src, dst in {"bench_2m": R + "_bench_2m_laplace.json", "bench_200k_one_rank_2threads": R + "_bench_200k_one_rank_2_host_threads.json",
                 "bench_trans_T_rhs16": R + "_bench_1m_laplace_transposed_rhs16.json", "bench_sym_rhs16": R + "_bench_1m_laplace_sym_one_triangle_rhs16.json"}.items():
    cp(os.path.join(F2, src + ".json"), dst)
pmc16 = os.path.join(ROOT, "gpurun_out", "r04pmc16")
if os.path.isdir(pmc16):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_kernel_table.py"), pmc16, "tile_gemm_wide16", "tile_gemm_tall16", "finish_sym16", "reduce_partials16", "gather_x16"],
                         capture_output=True, text=True, check=True).stdout
    with open(os.path.join(P, R + "_pmc_wide16_sweeps.txt"), "w") as f:
        f.write("# rocprofv3 --pmc passes (tools/sessions/r04_pmc_wide16.sh, taken BEFORE the slab-stride / finishing-pass changes of round 4) over the 16-wide sweeps of the\n"
                "# 1 M-point Laplace operator: sym16 = --symmetric one-triangle --rhs 16, wide16 = --rhs 16, trans16 = --trans T --rhs 16.  Mean counter values per launch;\n"
                "# SQ_* wave counters in quad-cycles summed over all waves, FETCH_SIZE / WRITE_SIZE in KB as reported (16-byte-per-lane streams: read bytes = 2 x FETCH_SIZE).\n")
        f.write(out)
    print("derived", R + "_pmc_wide16_sweeps.txt")
for name in ("r04w16", "r04tb"):
    d = os.path.join(ROOT, "gpurun_out", name)
    if os.path.isdir(d):
        lines = []
        for path in sorted(glob.glob(os.path.join(d, "*.json"))):
            try:
                j = json.loads(open(path).read().strip().splitlines()[-1])
                r = j["roofline"]
                lines.append("%-22s ms/step %8.3f  dominant interval %9.1f us  frac %.3f  %s" % (os.path.basename(path)[:-5], j["ms_per_step"], r["launch_us"], r["frac"], r["kernel"][:60]))
            except Exception as e:
                lines.append("%-22s unreadable (%s)" % (os.path.basename(path), e))
        with open(os.path.join(P, "r04_%s.txt" % {"r04w16": "wide16_after_changes", "r04tb": "tile_order_blocks_ab"}[name]), "w") as f:
            f.write("# bench.py --no-cpu-baseline --no-warm-build lines of tools/sessions/%s (1 M-point Laplace); *_xcd / *_blocks: HTOOL_TILE_ORDER\n" % {"r04w16": "r04_wide16_ab.sh", "r04tb": "r04_tile_blocks_ab.sh"}[name])
            f.write("\n".join(lines) + "\n")
        print("wrote summary of", name)
