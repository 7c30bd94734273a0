"""Which exchange the GPU-resident solver operator takes (htool_python_amd/solver.py:DeviceOperator): the in-library one only
when it runs on device buffers or the process group is gloo anyway -- never the host-staged path under an nccl group
(ADVICE round 3).  Reference: the distributed product of src/htool/distributed_operator/distributed_operator.hpp:23-65."""
import pytest


class _FakeOp:
    def __init__(self, kind, rccl):
        self._kind, self.has_rccl = kind, rccl

    def exchange_kind(self, mu=1):
        return self._kind


@pytest.mark.parametrize("kind,rccl,backend,expect", [
    (0, False, "nccl", True),     # one rank: nothing to exchange
    (1, True, "nccl", True),      # device all-gather hook (RCCL), zero-copy layout
    (2, True, "nccl", True),      # ... padded layout
    (3, False, "nccl", False),    # host-staged under an nccl group: the SliceGatherer's device all-gather stays
    (4, False, "nccl", False),
    (3, False, "gloo", True),     # ranks sharing a GPU (rehearsal): the library's host-staged exchange is the path under test
    (3, False, None, True),       # no process group at all
    (-1, False, "gloo", False),   # extra user terms: no in-library product
])
def test_device_operator_exchange_selection(built, monkeypatch, kind, rccl, backend, expect):
    import torch.distributed as tdist

    from htool_python_amd.solver import DeviceOperator

    monkeypatch.setattr(tdist, "is_initialized", lambda: backend is not None)
    monkeypatch.setattr(tdist, "get_backend", lambda *a, **k: backend)
    assert DeviceOperator._library_exchange_is_the_fast_one(_FakeOp(kind, rccl)) is expect
    assert DeviceOperator._library_exchange_is_the_fast_one(None) is False
