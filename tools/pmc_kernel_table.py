#!/usr/bin/env python
"""Per-kernel table of rocprofv3 --pmc passes: for every directory given, the mean counter values and mean duration of the
kernels whose name contains one of the patterns (one line per kernel and pass).

    python tools/pmc_kernel_table.py gpurun_out/r04pmc16 tile_gemm_wide16 tile_gemm_tall16 finish_sym16 > profiles/r04_pmc_wide16_sweeps.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def table(root, patterns):
    out = {}
    for path in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        run = os.path.basename(os.path.dirname(path))
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(dict)
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if not any(p in name for p in patterns):
                    continue
                short = name.split("(")[0].replace("void hm::", "")
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[short][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        out[run] = {k: ({c: sum(v) / len(v) for c, v in acc[k].items()}, sum(dur[k].values()) / len(dur[k]), len(dur[k])) for k in acc}
    return out


if __name__ == "__main__":
    root, patterns = sys.argv[1], sys.argv[2:]
    for run, kernels in table(root, patterns).items():
        for k, (ctr, us, n) in sorted(kernels.items()):
            print(f"{run:14s} {k[:70]:70s} {n:3d} launches, {us:9.1f} us  " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(ctr.items())))
