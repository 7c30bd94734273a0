import os

import numpy as np

SUM, MAX, MIN = "sum", "max", "min"


class _Comm:
    def __init__(self):
        self._pg_ready = False

    # -- identity (from the torchrun environment; a plain `python script.py` is a world of one)
    @property
    def size(self):
        return int(os.environ.get("WORLD_SIZE", "1"))

    @property
    def rank(self):
        return int(os.environ.get("RANK", "0"))

    def Get_size(self):
        return self.size

    def Get_rank(self):
        return self.rank

    def _dist(self):
        import torch.distributed as dist

        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.size)
        return dist

    def _host_group(self):
        """gloo group for host buffers (the default group may be nccl in GPU benches)."""
        dist = self._dist()
        if dist.get_backend() == "gloo":
            return None
        if not hasattr(self, "_gloo"):
            self._gloo = dist.new_group(backend="gloo")
        return self._gloo

    # -- RCCL: the exchange of the GPU-resident product inside the library (one process per GPU)
    def use_rccl(self):
        """Create the library-owned RCCL communicator behind this communicator (collective over all ranks).  Afterwards a
        DefaultApproximationBuilder built with it exchanges vector slices on device buffers (distributed_operator.matvec_device)
        and serves the replicated-vector API through RCCL too.  Needs one GPU per rank (RCCL cannot share a device)."""
        if getattr(self, "_rccl", None) is None:
            import Htool

            uid = Htool.rccl_unique_id() if self.rank == 0 else None
            import torch.distributed as tdist

            dist = tdist if tdist.is_initialized() else (self._dist() if self.size > 1 else None)
            if dist is not None and dist.get_backend() == "nccl":
                # the 128 bytes travel over the process group that is already there (a device tensor: no second, host-side
                # group has to be set up just for them)
                import torch

                box = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if self.rank == 0:
                    box.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
                dist.broadcast(box, src=0)
                uid = bytes(box.cpu().numpy().tobytes())
            else:
                uid = self.bcast(uid, root=0)
            self._rccl = Htool.RcclCommunicator(uid, self.rank, self.size)
        return self

    @property
    def _htool_comm_ptr(self):
        r = getattr(self, "_rccl", None)
        return None if r is None else r._htool_comm_ptr

    def Barrier(self):
        if self.size > 1:
            self._dist().barrier(group=self._host_group())

    barrier = Barrier

    def allreduce(self, value, op=SUM):
        if self.size == 1:
            return value
        import torch

        dist = self._dist()
        t = torch.tensor(np.asarray(value, dtype=np.float64))
        dist.all_reduce(t, op={SUM: dist.ReduceOp.SUM, MAX: dist.ReduceOp.MAX, MIN: dist.ReduceOp.MIN}[op], group=self._host_group())
        out = t.numpy()
        if np.isscalar(value) or np.ndim(value) == 0:
            return type(value)(out.item()) if isinstance(value, (int, float)) else out.item()
        return out

    def bcast(self, obj, root=0):
        if self.size == 1:
            return obj
        dist = self._dist()
        box = [obj]
        dist.broadcast_object_list(box, src=root, group=self._host_group())
        return box[0]

    # -- used by libhtool_mi355x through the pybind shim: all-gather of host byte slices
    def _htool_allgatherv(self, send, recv, counts, displs):
        if self.size == 1:
            recv[displs[0]:displs[0] + counts[0]] = send
            return
        import torch

        dist = self._dist()
        pad = max(counts)
        s = torch.zeros(pad, dtype=torch.uint8)
        s[: len(send)] = torch.from_numpy(np.ascontiguousarray(send))
        parts = [torch.empty(pad, dtype=torch.uint8) for _ in range(self.size)]
        dist.all_gather(parts, s, group=self._host_group())
        for p in range(self.size):
            recv[displs[p]:displs[p] + counts[p]] = parts[p][: counts[p]].numpy()


COMM_WORLD = _Comm()
Comm = _Comm
