"""MI355X-native H-matrix engine behind the `Htool` Python API (package root).

`torch` is imported before the extension on purpose: the PyTorch-ROCm wheel bundles its own HIP
runtime (libamdhip64.so.7 + libhsa-runtime64), and a process must only ever load one.  Importing
torch first makes libhtool_mi355x.so bind to that copy, so torch tensors (device memory, streams,
torch.distributed/RCCL) and the library share one runtime.  PyTorch is plumbing here, not compute.
"""
try:  # noqa: SIM105
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch-less deployments use the system ROCm runtime
    pass
