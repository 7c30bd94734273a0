"""Multi-process tests (one process per rank, as the reference runs its suite under mpirun -np k,
.github/workflows/CI.yml:138-145).  CPU: gloo, world size 2 and 3.  GPU: two ranks sharing the box's GPU."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, mode, timeout=240, worker="dist_worker.py"):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", worker), mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    for p in range(world):
        assert f"rank {p}/{world} ok" in r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_comm_shim_and_partition_logic_gloo(built, oracle, world):
    _launch(world, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_distributed_operator_gpu(built, oracle, world):
    _launch(world, "gpu")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_library_exchange_with_several_ranks_on_one_gpu(built, oracle, world):
    """htool_distributed_matvec_device / _matmat_device ITSELF with 2 and 3 ranks (gloo, one GPU): even and uneven partitions,
    1 and 3 columns, zero-copy and padded + compaction layouts, real and complex -- tests/dist_device_worker.py."""
    _launch(world, "device-exchange", timeout=400, worker="dist_device_worker.py")
