"""Vector-slice exchange of the row-distributed product: every rank contributes its slice of a
cluster-numbered vector, every rank receives the whole vector.

nccl (= RCCL over xGMI, one GPU per rank): one all-gather on device buffers -- `all_gather_into_tensor` on the
slices themselves when the partition is even, on zero-padded equal-size slices otherwise (RCCL/gloo all-gathers want
equal counts).  gloo: the same through host staging, so that several ranks can share one GPU in tests/rehearsals.
"""
import torch
import torch.distributed as dist


class SliceGatherer:
    def __init__(self, sizes, dtype, device, group=None):
        self.sizes, self.group = list(sizes), group
        self.world = len(self.sizes)
        self.equal = len(set(self.sizes)) == 1
        self.pad = max(self.sizes)
        self.offs = [sum(self.sizes[:p]) for p in range(self.world)]
        self.n = sum(self.sizes)
        self.nccl = dist.get_backend(group) == "nccl"
        self.full = torch.zeros(self.n, dtype=dtype, device=device)
        if self.nccl and not self.equal:
            self.x_pad = torch.zeros(self.pad, dtype=dtype, device=device)
            self.gathered = torch.zeros(self.world, self.pad, dtype=dtype, device=device)
        if not self.nccl:
            self.h_pad = torch.zeros(self.pad, dtype=dtype)
            self.h_parts = [torch.zeros(self.pad, dtype=dtype) for _ in range(self.world)]

    def __call__(self, x_local):
        """Returns the gathered vector (a persistent device buffer, overwritten by the next call)."""
        if self.nccl:
            if self.equal:
                dist.all_gather_into_tensor(self.full, x_local.contiguous(), group=self.group)
            else:
                self.x_pad[: x_local.numel()].copy_(x_local)
                dist.all_gather_into_tensor(self.gathered, self.x_pad, group=self.group)
                for p in range(self.world):
                    self.full[self.offs[p]: self.offs[p] + self.sizes[p]].copy_(self.gathered[p, : self.sizes[p]])
        else:
            self.h_pad[: x_local.numel()].copy_(x_local)
            dist.all_gather(self.h_parts, self.h_pad, group=self.group)
            self.full.copy_(torch.cat([self.h_parts[p][: self.sizes[p]] for p in range(self.world)]))
        return self.full


def all_reduce_sum(t, group=None):
    """In-place sum of a small device tensor over the ranks (device collective on nccl, host-staged on gloo)."""
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, group=group)
    else:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    return t
