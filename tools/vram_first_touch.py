"""How long do large device allocations take on this box, the first time and again?  (a fresh box: the first allocation of a VRAM
region has been seen to cost ~23 ms per GB; see NOTES.md)   python tools/vram_first_touch.py [GB]"""
import sys
import time

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
torch.cuda.init()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.time()
    a = torch.empty(int(gb * 1e9), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.time()
    del a
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"rep {rep}: alloc {gb:.0f} GB {t1 - t0:.3f} s, free {t2 - t1:.3f} s", flush=True)
    time.sleep(1.0 if rep == 0 else 6.0)
