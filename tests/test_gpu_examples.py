"""The example scripts run end to end on the GPU (they assert their own accuracy)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=180):
    env = dict(os.environ, MPLBACKEND="Agg", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("native", [False, True])
def test_use_hmatrix(built, tmp_path, native):
    out = _run([sys.executable, "examples/hmatrix_quickstart.py", "--plot", str(tmp_path / "h.png")] + (["--native"] if native else []))
    assert "matvec error" in out and (tmp_path / "h.png").exists()


@pytest.mark.parametrize("world", [1, 2])
def test_use_distributed_operator(built, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "examples/distributed_gmres.py"]
    out = _run(cmd)
    assert "solution error" in out
