"""The two data-parallel primitives of the device-resident build (csrc/device_scan.inc) against numpy: the ordered multi-counter
scan that hands out queue positions (block tree of src/htool/hmatrix/hmatrix_tree_builder.hpp:36 flattened on the device) and the
stable radix sort of (key, value) pairs (size classes of an ACA round, leaves by cluster node for the panel layout)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 7, 2047, 2048, 2049, 100_000, 3_000_001])
def test_ordered_scan_positions(built, n):
    import torch

    L = ctypes.CDLL(built[0])
    L.htool_last_error.restype = ctypes.c_char_p
    rng = np.random.RandomState(n % 97)
    counts = rng.randint(0, 5, size=(max(n, 1), 2)).astype(np.int32)
    counts[rng.rand(max(n, 1)) < 0.3] = 0  # many elements ask for nothing
    c = torch.from_numpy(counts).cuda()
    pos = torch.full((max(n, 1), 2), -1, dtype=torch.int64, device="cuda")
    tot = (ctypes.c_int64 * 2)()
    assert L.htool_debug_scan_positions(ctypes.c_void_p(c.data_ptr()), ctypes.c_int64(n), ctypes.c_void_p(pos.data_ptr()), tot) == 0, L.htool_last_error()
    want = np.cumsum(counts[:n].astype(np.int64), axis=0) - counts[:n]
    assert np.array_equal(pos.cpu().numpy()[:n], want)
    assert [tot[0], tot[1]] == (counts[:n].astype(np.int64).sum(axis=0).tolist() if n else [0, 0])


@pytest.mark.parametrize("n,bits,key_range", [(0, 8, 4), (1, 3, 8), (5000, 3, 7), (2048, 8, 256), (2049, 20, 1 << 20), (1_000_003, 19, 300_000), (4_000_000, 30, 1 << 30),
                                              (300_000, 32, None)])
def test_stable_radix_sort_of_pairs(built, n, bits, key_range):
    import torch

    L = ctypes.CDLL(built[0])
    L.htool_last_error.restype = ctypes.c_char_p
    rng = np.random.RandomState(bits)
    if key_range is None:
        keys = rng.randint(0, 1 << 32, size=max(n, 1), dtype=np.uint64).astype(np.uint32)
    else:
        keys = rng.randint(0, key_range, size=max(n, 1)).astype(np.uint32)  # many ties: the sort has to be STABLE
    vals = np.arange(max(n, 1), dtype=np.uint32)
    k, v = torch.from_numpy(keys.view(np.int32)).cuda(), torch.from_numpy(vals.view(np.int32)).cuda()
    assert L.htool_debug_sort_pairs(ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(v.data_ptr()), ctypes.c_int64(n), bits) == 0, L.htool_last_error()
    order = np.argsort(keys[:n], kind="stable")
    assert np.array_equal(k.cpu().numpy().view(np.uint32)[:n], keys[:n][order])
    assert np.array_equal(v.cpu().numpy().view(np.uint32)[:n], vals[:n][order])
