#!/usr/bin/env python
"""bench.py -- H-matvec GB/s (+ build seconds) for an N-point 3-D Laplace kernel on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json north_star / configs[3]): the 1 000 000-point 3-D Laplace single-layer
operator 1/(4 pi r), points uniform in the unit ball (seed 0, construction of the reference's
example/create_geometry.py:13-22), eta=10, eps=1e-3, binary PCA-regular cluster tree, fp64.
A step is ONE H-matrix-vector product y = H x with x and y resident in HBM (user numbering in and
out at N=1, exactly what `hmatrix * x` computes).  With N>1 the rows are split over the GPUs by the
depth-1 partition of the cluster tree (DefaultApproximationBuilder's decomposition); a step is then an
RCCL all-gather of the x slices followed by the local product (strong scaling: total work fixed).
value = algorithmic bytes of all ranks / time; the roofline object prices the dominant
kernel (phase B, tile_gemv_wide) with HIP events recorded on its stream inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md; measured copy ceiling 6290)


def kernel_source_sha1():
    """Fingerprint of every source that decides what the dominant kernel reads -- the kernels, the panel layout and packing, the
    launch classes: committed PMC measurements are only quoted for the code they were taken on."""
    import hashlib

    h = hashlib.sha1()
    for name in ("product_kernels.inc", "product_mfma.inc", "device_tables.inc", "pack_kernels.inc", "layout.cpp", "blocktree.cpp", "device.hip", "device_build2.inc", "device_scan.inc"):
        with open(os.path.join(ROOT, "htool_python_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic(args, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command line
    (profiles/), or None when no pass exists for this workload OR the kernels have changed since the passes were taken
    (the file records the fingerprint of the kernel source it was measured on)."""
    if not (n == 1_000_000 and args.kernel == "laplace" and args.eps == 1e-3 and args.eta == 10.0 and args.leaf == 100 and args.gpus == 1 and args.rhs == 1):
        return None, None
    for name in ("r04_pmc_hbm_traffic_1m_laplace.json", "r03_pmc_hbm_traffic_1m_laplace.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_source_sha1") == kernel_source_sha1():
            return d.get("tile_gemv_wide_hbm_bytes_per_launch"), name
    return None, None


def cpu_full_operator(n_points=100_000, eps=1e-4, eta=10.0, leaf=100, budget_s=8.0):
    """BASELINE config C2 (100 000-point Laplace, eps 1e-4: the whole operator fits the host, 8 GB): WHOLE build and WHOLE
    product on the CPU (oracle/hmat_oracle.cpp, OpenMP over blocks / leaves) next to this engine on the same instance."""
    import numpy as np
    import torch

    import Htool
    from htool_python_amd.workloads import algorithmic_bytes, points_in_sphere
    from oracle import oracle as O

    pts = points_in_sphere(n_points, seed=0)
    t0 = time.perf_counter()
    oc = O.Cluster(pts, max_leaf=leaf)
    OH = O.HMatrix(oc, oc, O.K_LAPLACE, eps=eps, eta=eta)
    t_cpu_build = time.perf_counter() - t0
    x = np.random.RandomState(0).rand(n_points)
    t_all = []
    t_start = time.time()
    while len(t_all) < 3 or (time.time() - t_start < budget_s and len(t_all) < 20):
        t0 = time.perf_counter()
        y_cpu = OH.matvec(x)
        t_all.append(time.perf_counter() - t0)
    t_cpu = sorted(t_all)[len(t_all) // 2]
    ab_cpu = algorithmic_bytes(OH.leaves, n_points, n_points, 8)["total"]
    t0 = time.perf_counter()
    cb = Htool.ClusterTreeBuilder()
    cb.set_maximal_leaf_size(leaf)
    cl = cb.create_cluster_tree(pts, 2)
    Hs = Htool.HMatrixTreeBuilder(eps, eta, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts), cl, cl)
    torch.cuda.synchronize()
    t_gpu_build = time.perf_counter() - t0
    gpu_leaves = Hs.leaves()
    ab_gpu = algorithmic_bytes(gpu_leaves, n_points, n_points, 8)["total"]
    xd, yd = torch.from_numpy(x).cuda(), torch.zeros(n_points, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        Hs.matvec_device(xd.data_ptr(), yd.data_ptr(), 0, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        Hs.matvec_device(xd.data_ptr(), yd.data_ptr(), 0, st)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / 20
    y_gpu = yd.cpu().numpy()
    del Hs
    Htool.release_workspace()
    return {"workload": f"BASELINE C2: {n_points}-point 3D Laplace, eps={eps:g}, eta={eta:g}, leaf={leaf} -- the whole operator, whole build, on the CPU",
            "cpu_build_s": t_cpu_build, "cpu_matvec_GBps": ab_cpu / t_cpu / 1e9, "cpu_matvec_ms": t_cpu * 1e3, "cpu_algorithmic_GB": ab_cpu / 1e9,
            "threads": O.num_threads(), "passes": len(t_all),
            "gpu_build_s": t_gpu_build, "gpu_matvec_GBps": ab_gpu / t_gpu / 1e9, "gpu_matvec_ms": t_gpu * 1e3,
            "same_leaf_count": bool(len(OH.leaves) == len(gpu_leaves)),
            "rel_diff_cpu_gpu_product": float(np.linalg.norm(y_gpu - y_cpu) / np.linalg.norm(y_cpu))}


def cpu_baseline(H, leaves, n_rows, n_source, elem_bytes, budget_s=12.0):
    """CPU leaf loop (oracle, OpenMP) on a bounded random sample of this operator's own leaves."""
    import numpy as np

    from oracle import oracle as O

    O.set_num_threads(O.usable_cpus())
    rng = np.random.RandomState(0)
    L = np.asarray(leaves, dtype=np.int64)
    size = np.where(L[:, 4] < 0, L[:, 1] * L[:, 3], L[:, 4] * (L[:, 1] + L[:, 3])) * elem_bytes
    order = rng.permutation(len(L))
    cap_bytes, cap_leaves = 8e9, 150000
    pick, tot = [], 0
    for i in order:
        if L[i, 4] == 0:
            continue
        if tot + size[i] > cap_bytes or len(pick) >= cap_leaves:
            break
        pick.append(i)
        tot += int(size[i])
    sel, offs, panels = O.leaf_sample_panels(H, pick)
    # the full-size CPU loop pays (threads x N) for its thread-private y once per 85 GB of panels; keep
    # that overhead proportionate for the sample by folding the sampled leaves' rows into a window
    frac = tot / float(size.sum())
    win = int(max(4 * frac * n_rows, sel[:, 1].max(), 4096))
    sel = sel.copy()
    sel[:, 0] = sel[:, 0] % (win - sel[:, 1] + 1)
    n_rows = win
    xp = rng.rand(n_source)
    t_all, reps = [], 0
    t_start = time.time()
    while reps < 3 or (time.time() - t_start < budget_s and reps < 50):
        t0 = time.perf_counter()
        O.leaf_loop(sel, offs, panels, n_rows if n_rows > 0 else 1, xp)
        t_all.append(time.perf_counter() - t0)
        reps += 1
    t_med = sorted(t_all)[len(t_all) // 2]
    del panels
    return {
        "full_operator": cpu_full_operator(),
        "value": tot / t_med / 1e9,
        "unit": "GB/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": f"CPU leaf loop (oracle/hmat_oracle.cpp, OpenMP) over {len(pick)} randomly drawn leaves of this operator "
                  f"({tot / 1e9:.2f} GB of its {size.sum() / 1e9:.1f} GB of panels, downloaded from HBM), median of {reps} passes; "
                  "the reference's own C++/MPI path (lib/htool) is not in the container",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", dest="n", type=int, default=1_000_000, help="number of points")
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--eta", type=float, default=10.0)
    ap.add_argument("--leaf", type=int, default=100, help="maximal_leaf_size of the cluster tree")
    ap.add_argument("--kernel", default="laplace", choices=["laplace", "inv_delta", "helmholtz"])
    ap.add_argument("--kappa", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: run the multi-rank code path (process group, slice exchange, "
                                                               "cluster-numbered local product) even with a single rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the measured path; gloo stages the exchange through the host so that the "
                         "multi-rank logic can be rehearsed with several ranks on ONE GPU (not a benchmark)")
    ap.add_argument("--recompress", action="store_true", help="run Htool.recompression (device SVD recompression of the low-rank leaves) after the build")
    ap.add_argument("--rhs", type=int, default=1, help="right-hand sides per step (H @ X, one sweep of the panels per 8 columns); 1 = the headline matvec")
    ap.add_argument("--gmres", type=int, default=0, help="BASELINE config 5: instead of bare products, a step is ONE GMRES iteration (restart = this value) "
                                                        "on (shift I + H) with device-resident Krylov vectors")
    ap.add_argument("--shift", type=float, default=0.0, help="diagonal shift of the --gmres system (0: N/50, keeps the system well posed)")
    ap.add_argument("--symmetric", choices=["full", "one-triangle"], default=None,
                    help="build with symmetry 'S' / UPLO 'L' (single GPU only): 'full' stores both triangles (the default engine "
                         "layout), 'one-triangle' stores the lower triangle only and uses every leaf twice in a fused sweep")
    ap.add_argument("--no-phase-timing", action="store_true", help="do not record the per-phase HIP events (the roofline object is then empty); "
                                                                    "lets the library replay repeated products as a hipGraph")
    ap.add_argument("--trans", default="N", choices=["N", "T", "C"], help="time the transposed product y = H^T x (H^H x) instead (single GPU; not the headline metric: "
                                                                          "no per-phase events, the roofline object stays empty)")
    ap.add_argument("--no-warm-build", action="store_true", help="build the operator once only: build_s is then build_cold_s, the first build of the process "
                                                                  "(code objects, streams, first touch of the memory); default: the operator is built twice and both "
                                                                  "times are reported")
    ap.add_argument("--check", action="store_true", help="(kept for compatibility: the error against sampled exact rows is always reported)")
    args = ap.parse_args()
    # stdout carries exactly ONE line (the JSON): libraries that chat on stdout (RCCL prints a version banner when a
    # communicator is created) are sent to stderr by pointing fd 1 at fd 2 until the result is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.backend == "gloo":  # rehearsal: ranks may share a GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # every torch op and every product of this process goes to ONE explicit stream (not the legacy default stream): ordering is
    # then plain stream order, and the library may replay the captured launches of a repeated product as a hipGraph
    torch.cuda.set_stream(torch.cuda.Stream())
    import torch.distributed as dist

    dist_mode = world > 1 or args.force_dist
    if dist_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    import Htool
    from htool_python_amd.workloads import algorithmic_bytes, points_in_sphere, usable_cpus

    Htool.set_device(local_rank)       # (also warms the library up: code objects of all kernels, build streams)
    warm_up_s = Htool.last_warm_up_seconds()
    # host threads of this rank (what is left on the host: tiles, per-node layout, table assembly; the cluster tree and the
    # block tree are built on the GPU): its share of the CPUs, stated in the line (torchrun exports OMP_NUM_THREADS=1 for
    # multi-rank launches, and a node-wide default would oversubscribe 8 ranks)
    host_threads = max(1, min(16, usable_cpus() // world))
    if os.environ.get("HTOOL_BENCH_THREADS"):
        host_threads = max(1, int(os.environ["HTOOL_BENCH_THREADS"]))
    Htool.set_num_threads(host_threads)
    is_complex = args.kernel == "helmholtz"
    elem = 16 if is_complex else 8
    dtype = torch.complex128 if is_complex else torch.float64
    n = args.n

    # the library's own account of the build (INFO line "native build timing: ...") goes into the JSON line
    import logging

    build_log = []

    class _Keep(logging.Handler):
        def emit(self, record):
            msg = record.getMessage()
            if "native build timing" in msg or "build timeline" in msg:
                build_log.append(msg)
            elif "H-matrix built" in msg and build_log:  # (the C-ABI call as a whole: tiles, the native build, its clean-up)
                build_log[-1] += "; " + msg

    logging.getLogger("Htool").addHandler(_Keep())
    logging.getLogger("Htool").setLevel(logging.DEBUG)  # (the build's stage@seconds timeline is a DEBUG line)
    pts = points_in_sphere(n, seed=0)
    # the cluster tree (built on the GPU, csrc/cluster_device.hip; every rank builds the same tree on its own GPU) -- like the
    # operator it is built twice: cluster_tree_cold_s is the first tree of the process (workspace allocated), cluster_tree_s the second
    t0 = time.time()
    cb = Htool.ClusterTreeBuilder()
    cb.set_maximal_leaf_size(args.leaf)
    cluster = cb.create_cluster_tree(pts, 2, size_of_partition=world)
    t_cluster_cold = time.time() - t0
    t_cluster = t_cluster_cold
    if not args.no_warm_build:
        del cluster
        t0 = time.time()
        cluster = cb.create_cluster_tree(pts, 2, size_of_partition=world)
        t_cluster = time.time() - t0
    param = {"laplace": 0.0, "inv_delta": 0.1, "helmholtz": args.kappa}[args.kernel]
    if is_complex:
        gen = Htool.ComplexNativeGenerator(args.kernel, pts, pts, param)
        builder = Htool.ComplexHMatrixTreeBuilder(args.eps, args.eta, "S" if args.symmetric else "N", "L" if args.symmetric else "N")
    else:
        gen = Htool.NativeGenerator(args.kernel, pts, pts, param)
        builder = Htool.HMatrixTreeBuilder(args.eps, args.eta, "S" if args.symmetric else "N", "L" if args.symmetric else "N")
    assert args.trans == "N" or not dist_mode, "--trans is a single-GPU option (the distributed operator is 'N'-only, as the reference's)"
    if args.symmetric:
        assert not dist_mode, "--symmetric is a single-GPU option (a row partition is not a symmetric operator)"
        builder.set_symmetric_storage(args.symmetric == "one-triangle")
    comm, rccl_error = None, None
    if dist_mode:
        # the reference's decomposition through its own entry point (DefaultApproximationBuilder, utility.hpp:26): rank p builds
        # rows(partition p) x all columns; the communicator carries a library-owned RCCL handle, so the exchange of every
        # product runs inside the library on device buffers (htool_distributed_matvec_device).  --backend gloo (rehearsal, ranks
        # sharing a GPU): the SAME library call, its all-gather staged through the host by the library (htool_comm without
        # allgather_device)
        import mpi4py

        comm = mpi4py.MPI.COMM_WORLD
        if args.backend == "nccl":
            try:
                comm.use_rccl()
            except Exception as e:  # reported in the JSON line ("exchange"); the exchange then goes through torch.distributed (also RCCL)
                rccl_error = repr(e)

    def timed_build():
        build_log.clear()
        torch.cuda.synchronize()
        t0 = time.time()
        if dist_mode:
            approx = Htool.DefaultApproximationBuilder(gen, cluster, cluster, builder, comm)
            op_, H_ = approx.distributed_operator, approx.hmatrix
        else:
            approx, op_ = None, None
            H_ = builder.build(gen, cluster, cluster, -1)
        t_call = time.time() - t0
        torch.cuda.synchronize()
        t_all = time.time() - t0
        return approx, op_, H_, t_all, ((build_log[-1] + f"; builder.build() returned after {t_call:.3f} s, device idle after {t_all:.3f} s") if build_log else None)

    # FIRST build of the process, nothing warmed or pre-conditioned: code objects of the build kernels are loaded, streams created,
    # every byte of VRAM the build uses is handed out (and wiped) by the driver for the first time -- build_cold_s
    approx, dist_op, H, t_build_cold, cold_breakdown = timed_build()
    t_build, build_breakdown = t_build_cold, cold_breakdown
    vram_prep = None
    if not args.no_warm_build:
        # SECOND build of the process = build_s: the steady state of a process that builds operators repeatedly.  The first
        # operator is destroyed; the library keeps its temporary ACA arena in the workspace cache (as it does for any later build),
        # the panels go back to the driver, which wipes freed memory in the background (~45 ms per GB) while new allocations wait
        # for it (tools/vram_first_touch.py) -- so the build waits, untimed, for that wipe instead of racing it.  (Round 2
        # allocated 92 % of the free memory here; nothing is allocated any more: only what the first build itself touched is reused.)
        freed_GB = H.stats()["hbm_bytes"] / 1e9
        del approx, dist_op, H
        torch.cuda.synchronize()
        wait_s = 1.0 + 0.06 * freed_GB
        time.sleep(wait_s)
        vram_prep = {"freed_by_first_build_GB": freed_GB, "wait_s": wait_s, "allocated_for_preconditioning_GB": 0.0}
        approx, dist_op, H, t_build, build_breakdown = timed_build()
    t_recompress = None
    if args.recompress:
        t0 = time.time()
        Htool.recompression(H)
        torch.cuda.synchronize()
        t_recompress = time.time() - t0
    H.set_phase_timing(not args.no_phase_timing)  # HIP events around every launch of a product (the roofline object needs them)
    leaves = H.leaves()
    n_rows = H.shape[0]
    ab = algorithmic_bytes(leaves, n, n_rows, elem)
    stats = H.stats()

    stream = torch.cuda.current_stream().cuda_stream
    gen_t = torch.Generator(device="cpu").manual_seed(1234 + rank)
    if not dist_mode:
        x = torch.rand(n, dtype=torch.float64, generator=gen_t)
        if is_complex:
            x = torch.complex(x, torch.rand(n, dtype=torch.float64, generator=gen_t))
        x = x.cuda()
        y = torch.zeros(n, dtype=dtype, device="cuda")

        if args.rhs > 1:
            x = torch.rand(args.rhs, n, dtype=torch.float64, generator=gen_t).to(dtype).cuda()
            y = torch.zeros(args.rhs, n, dtype=dtype, device="cuda")

        def step():
            if args.trans != "N":
                H.matmat_device_trans(args.trans, x.data_ptr(), n, y.data_ptr(), n, args.rhs, 0, stream)
            elif args.rhs > 1:
                H.matmat_device(x.data_ptr(), n, y.data_ptr(), n, args.rhs, 0, stream)
            else:
                H.matvec_device(x.data_ptr(), y.data_ptr(), 0, stream)
    else:
        local = cluster.get_cluster_on_partition(rank)
        sizes = [cluster.get_cluster_on_partition(p).get_size() for p in range(world)]
        x_local = torch.rand(local.get_size(), dtype=torch.float64, generator=gen_t)
        if is_complex:
            x_local = torch.complex(x_local, torch.rand(local.get_size(), dtype=torch.float64, generator=gen_t))
        x_local = x_local.cuda()
        y = torch.zeros(n_rows, dtype=dtype, device="cuda")
        from htool_python_amd.comm import SliceGatherer

        gather = SliceGatherer(sizes, dtype, "cuda")

        in_library = dist_op is not None and dist_op.exchange_kind(1) >= 0 and (dist_op.has_rccl or args.backend == "gloo")
        if in_library:
            def step():
                # one library call: all-gather of the x slices (RCCL over xGMI; host-staged in a gloo rehearsal) + compaction +
                # the local product, on one stream
                dist_op.matvec_device(x_local.data_ptr(), y.data_ptr(), stream)
        else:
            def step():
                # the library communicator could not be set up: torch.distributed all-gather (also RCCL), then the local product
                x_full = gather(x_local)
                H.matvec_device(x_full.data_ptr(), y.data_ptr(), 1, stream)

    gmres_info = None
    if args.gmres > 0:
        from htool_python_amd.krylov import gmres
        from htool_python_amd.solver import DeviceOperator
        from htool_python_amd.workloads import gmres_shift

        part = [(cluster.get_cluster_on_partition(p).get_offset(), cluster.get_cluster_on_partition(p).get_size()) for p in range(world)] if dist_mode else None
        plain = DeviceOperator(H, part, rank, None, 0.0, dist_op=dist_op)
        if args.shift != 0.0:
            shift, norm_est = args.shift, None
        else:  # a fixed fraction of |A| (power iterations): well posed, but the Krylov method needs its iterations
            shift, norm_est = gmres_shift(plain.apply, plain.size, dtype, reduce=plain.reduce if dist_mode else None)
        op = DeviceOperator(H, part, rank, None, shift, dist_op=dist_op)
        x_ref = torch.rand(op.size, dtype=torch.float64, generator=gen_t).to(dtype).cuda()
        b_local = op.apply(x_ref)  # use_ddm_solver.py:60-61: b = A x_ref
        red = op.reduce if dist_mode else None
        args.steps = args.gmres

        def run_gmres():
            return gmres(op.apply, b_local, tol=0.0, restart=args.gmres, max_it=args.gmres, reduce=red)

        run_gmres()  # warm-up cycle (allocations, RCCL channels)
        step = None
    for _ in range(args.warmup if step is not None else 0):
        step()
    torch.cuda.synchronize()
    if dist_mode:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if step is None:
        xs_, gmres_info = run_gmres()
        gmres_info["x"] = xs_
    else:
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize()
    if dist_mode:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_ph, ph = H.phase_times_us()
    exchange_us = None
    if dist_mode and step is not None:
        # the exchange on its own (reported separately; it is inside the timed steps as well): the same steps
        # minus the same number of local products on an already gathered vector
        x_full_once = gather(x_local).clone()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            H.matvec_device(x_full_once.data_ptr(), y.data_ptr(), 1, stream)
        torch.cuda.synchronize()
        local_only = (time.perf_counter() - t1) / args.steps
        exchange_us = max(0.0, dt / args.steps - local_only) * 1e6
        step()  # (y holds the distributed product again)
        torch.cuda.synchronize()

    tot_bytes = float(ab["total"])
    # correctness figure of every run: the product against exact rows of the dense operator (256 sampled rows; column 0 when
    # several right-hand sides are multiplied)
    rel_err = None
    if step is not None:
        from oracle import oracle as O

        kind = {"laplace": 1, "inv_delta": 0, "helmholtz": 2}[args.kernel]
        if not dist_mode:
            rows = np.arange(0, n, max(1, n // 256))
            xx, yy = x.cpu().numpy(), y.cpu().numpy()
            if args.rhs > 1:
                xx, yy = xx[0], yy[0]
            if args.trans == "C":  # the kernels are symmetric (A^T = A): A^H x = conj(A conj(x))
                ye = np.conj(O.dense_matvec(kind, pts, pts, np.conj(xx), param, rows=rows))
            else:
                ye = O.dense_matvec(kind, pts, pts, xx, param, rows=rows)
            rel_err = float(np.linalg.norm(yy[rows] - ye) / np.linalg.norm(ye))
        else:
            # every rank checks rows of its own partition against the exact operator applied to the gathered x
            perm = np.asarray(cluster.get_permutation())
            x_user = np.zeros(n, dtype=np.complex128 if is_complex else np.float64)
            x_user[perm] = gather(x_local).cpu().numpy()
            off = local.get_offset()
            idx = np.arange(0, local.get_size(), max(1, local.get_size() // max(1, 256 // world)))
            ye = O.dense_matvec(kind, pts, pts, x_user, param, rows=perm[off + idx])
            dev = "cuda" if args.backend == "nccl" else "cpu"
            e2 = torch.tensor([float(np.linalg.norm(y.cpu().numpy()[idx] - ye) ** 2), float(np.linalg.norm(ye) ** 2)], dtype=torch.float64, device=dev)
            dist.all_reduce(e2)
            rel_err = float(torch.sqrt(e2[0] / e2[1]))
    elif gmres_info is not None:
        pass  # (the GMRES mode reports its true residual below)
    per_rank = None
    if dist_mode:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        mine = torch.tensor([tot_bytes, sum(ph) if n_ph else 0.0, exchange_us or 0.0, float(t_build), float(t_build_cold), float(t_cluster), float(t_cluster_cold), float(host_threads)],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"algorithmic_GB": float(v[0]) / 1e9, "product_us": float(v[1]), "exchange_us": float(v[2]), "build_s": float(v[3]), "build_cold_s": float(v[4]),
                     "cluster_tree_s": float(v[5]), "cluster_tree_cold_s": float(v[6]), "setup_s": float(v[3]) + float(v[5]), "host_threads": int(v[7])} for v in allr]
        t = torch.tensor([dt, tot_bytes, float(t_build), float(t_build_cold), float(t_cluster), float(t_cluster_cold)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, tot_bytes, t_build, t_build_cold, t_cluster, t_cluster_cold = float(tmax[0]), float(t[1]), float(tmax[2]), float(tmax[3]), float(tmax[4]), float(tmax[5])
    ms_per_step = dt / args.steps * 1e3
    value = tot_bytes / (dt / args.steps) / 1e9

    out = {
        "metric": "h_matvec_GBps",
        "value": value,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "c128" if is_complex else "f64",
        "data": "synthetic",
        "backend": ("rccl" if args.backend == "nccl" else "gloo-host-staged (rehearsal, not a benchmark)") if world > 1 else None,
        "config": {
            "workload": f"{n}-point 3D {args.kernel} H-matrix matvec (BASELINE configs[3] operator{' on one GPU' if world == 1 else ', row-cluster split'}), "
                        f"eta={args.eta:g}, eps={args.eps:g}, leaf={args.leaf}, ACA on device{' + SVD recompression' if args.recompress else ''}, points seed 0 unit ball",
            "n_points": n, "eps": args.eps, "eta": args.eta, "leaf": args.leaf, "kernel": args.kernel,
            "parallelism": f"rows{world}" if world > 1 else "single",
        },
        # build_cold_s: the FIRST build of the process, nothing warmed (code objects, streams, first touch of every VRAM page, the
        # driver's wipe of whatever ran on the GPU before).  build_s: the SECOND build of the same operator in the same process
        # (the first one destroyed, its panels' wipe waited for untimed, the library's workspace cache holding the ACA arena) --
        # the steady state of a process that builds operators repeatedly; equal to build_cold_s under --no-warm-build.
        # setup_s / setup_cold_s add the cluster tree (host), which every build needs in front of it.
        "build_s": t_build,
        "build_cold_s": t_build_cold,
        "cluster_tree_s": t_cluster,
        "cluster_tree_cold_s": t_cluster_cold,
        "setup_s": t_cluster + t_build,
        "setup_cold_s": t_cluster_cold + t_build_cold,
        "warm_up_s": warm_up_s,   # Htool.set_device(): code objects of all kernels loaded, build streams created (before anything is timed)
        "host_threads": host_threads,
        "build_warmed": not args.no_warm_build,
        "vram_preconditioning": vram_prep,
        "build_breakdown": build_breakdown,
        "build_cold_breakdown": cold_breakdown,
        "recompression_s": t_recompress,
        "algorithmic_GB": tot_bytes / 1e9,
        "rhs_per_step": args.rhs,
        "trans": args.trans,
        "symmetric_storage": args.symmetric,
        "per_rank": per_rank,   # multi-GPU: bytes, product time (sum of its kernels, HIP events) and exchange time of every rank
        "rel_err_sampled_rows": rel_err,
        "exchange": None if not dist_mode else (
            ("htool_distributed_matvec_device: ncclAllGather of the x slices + compaction + local product inside the library, one stream" if dist_op.has_rccl else
             "htool_distributed_matvec_device: inside the library, all-gather staged through the host (gloo rehearsal: ranks share a GPU; not a benchmark)")
            + f" [exchange kind {dist_op.exchange_kind(1)}]"
            if in_library else f"torch.distributed all_gather_into_tensor (RCCL) + local product; library communicator unavailable: {rccl_error}"),
    }
    if gmres_info is not None:
        # true residual and solution error of the timed solve (use_ddm_solver.py:60-61, tests/test_ddm_solver.py:659-660)
        xs = gmres_info.pop("x")
        res = gmres_info["residuals"]
        r_true = b_local - op.apply(xs)
        sq = torch.stack([torch.sum(torch.abs(r_true) ** 2), torch.sum(torch.abs(b_local) ** 2), torch.sum(torch.abs(xs - x_ref) ** 2), torch.sum(torch.abs(x_ref) ** 2)]).to(torch.float64)
        if dist_mode:
            op.reduce(sq)
        sq = sq.cpu().numpy()
        hit = [i + 1 for i, v in enumerate(res) if v <= 1e-6]
        out["gmres"] = {"iterations": gmres_info["iterations"], "s_per_iteration": dt / max(gmres_info["iterations"], 1),
                        "iterations_to_1e-6": hit[0] if hit else None,
                        "relative_residuals": [res[i] for i in (0, len(res) // 2, -1)],
                        "true_relative_residual": float(np.sqrt(sq[0] / sq[1])), "solution_error": float(np.sqrt(sq[2] / sq[3])),
                        "shift": shift, "operator_norm_estimate": norm_est,
                        "system": "(shift I + H) x = b with b = (shift I + H) x_ref, shift = 8e-3 |H| (power iterations): unpreconditioned GMRES needs ~40 iterations for 1e-6",
                        "note": "step = one GMRES iteration (1 product + CGS2 orthogonalisation), restart = number of steps, device-resident Krylov vectors"}
    if rank == 0:
        t_b = ph[3] * 1e-6 if n_ph else None
        fused = args.symmetric == "one-triangle"
        # one-triangle storage: the last event interval holds the fused sweep over the U / dense panels AND the second,
        # transposed, read of the V panels
        bytes_b = ab["phase_b"] + (ab["phase_a"] if fused else 0)
        traffic, traffic_file = (None, None) if args.symmetric else pmc_traffic(args, n)
        out["roofline"] = {
            "bound": "hbm",
            "kernel": (("tile_gemm_wide16_sym + tile_gemm_tall16_transposed" if args.rhs > 8 else "tile_gemv_wide_sym + tile_gemv_tall_transposed") + " (fused sweep of the stored triangle)") if fused
                      else ("tile_gemm_wide16 (phase B: U and dense panels, 16 right-hand sides on the matrix cores)" if args.rhs > 8 else "tile_gemv_wide (phase B: U and dense panels)"),
            "achieved": (bytes_b / t_b / 1e9) if t_b else None,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": (bytes_b / t_b / 1e9 / HBM_PEAK_GBPS) if t_b else None,
            # HBM bytes per launch from the COMMITTED rocprofv3 --pmc passes of this command line (rocprof cannot run inside this
            # process): quoted only while the kernel sources still have the fingerprint the passes were taken on
            "traffic": traffic,
            "traffic_source": None if traffic is None else f"profiles/{traffic_file} (FETCH_SIZE x 2 + WRITE_SIZE of separate --pmc passes; kernel sources sha1 {kernel_source_sha1()[:12]})",
            "launch_us": ph[3] if n_ph else None,
            "algorithmic_bytes_per_launch": bytes_b,
            "launches_averaged": n_ph,
            "other_kernels_us": {"x_gather": ph[0], "phase_a_tile_gemv_tall": ph[1], "phase_a2_tile_gemv_tall": ph[2]} if n_ph else None,
            "phase_a_achieved": (ab["phase_a"] / (ph[1] * 1e-6) / 1e9) if n_ph and ph[1] > 0 else None,
        }
        if args.trans != "N":
            # the transposed product records no per-phase events: its launches as a whole (x gather -> wide sweep, transposed use
            # only -> sums of the partials -> V^T pass -> finish), priced with the wall time of a step on the product's stream
            out["roofline"] = {"bound": "hbm", "kernel": ("tile_gemm_wide16_sym<DIRECT = false> + tile_gemm_tall16_transposed" if args.rhs > 8 else "tile_gemv_wide_sym<DIRECT = false> + tile_gemv_tall_transposed") + " (whole transposed product, all launches)",
                               "achieved": value, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": value / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                               "launch_us": ms_per_step * 1e3, "algorithmic_bytes_per_launch": tot_bytes, "launches_averaged": args.steps, "other_kernels_us": None, "phase_a_achieved": None}
        out["hmatrix"] = {"n_dense": stats["n_dense"], "n_low_rank": stats["n_low_rank"], "max_rank": stats["max_rank"],
                          "mean_rank": stats["sum_rank"] / max(stats["n_low_rank"], 1), "hbm_resident_GB": stats["hbm_bytes"] / 1e9}
        if world == 1 and not args.no_cpu_baseline and not is_complex:
            out["cpu_baseline"] = cpu_baseline(H, leaves, n, n, elem)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist_mode:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
