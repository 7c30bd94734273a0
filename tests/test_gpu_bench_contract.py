"""bench.py keeps its contract: exactly one JSON line on stdout with the agreed keys (single rank, and two ranks
sharing the GPU through the gloo rehearsal backend)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def _run(cmd):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE line, nothing else on stdout (RCCL's banner included)
    return json.loads(lines[0])


def test_single_rank_line(built):
    d = _run([sys.executable, "bench.py", "--points", "60000", "--steps", "4", "--warmup", "2"])
    assert KEYS <= set(d) and d["metric"] == "h_matvec_GBps" and d["unit"] == "GB/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["algorithmic_GB"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["launches_averaged"] >= 4 and rf["launch_us"] > 0
    # round 4: the set-up figures say what they are (second / first tree and build of the process, library warm-up before them)
    assert 0 < d["cluster_tree_s"] <= d["cluster_tree_cold_s"] + 0.05 and abs(d["setup_s"] - (d["cluster_tree_s"] + d["build_s"])) < 1e-9
    assert d["warm_up_s"] >= 0 and d["host_threads"] >= 1 and "timeline" in (d["build_breakdown"] or "")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "GB/s" and cb["value"] > 0 and cb["cores"] >= 1 and "leaves" in cb["sample"]


def test_two_rank_line_rehearsal(built):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
              "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "60001", "--backend", "gloo", "--check"])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "rows2"
    assert d["rel_err_sampled_rows"] < 1e-3
    assert len(d["per_rank"]) == 2 and all(r["host_threads"] >= 1 and r["cluster_tree_s"] > 0 and abs(r["setup_s"] - (r["cluster_tree_s"] + r["build_s"])) < 1e-9 for r in d["per_rank"])
