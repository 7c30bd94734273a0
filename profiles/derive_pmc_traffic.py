"""Derive HBM bytes per launch from the two rocprofv3 PMC passes committed next to this file.

    python profiles/derive_pmc_traffic.py profiles/r02_pmc_fetch_size_counter_collection.csv \
        profiles/r02_pmc_write_size_counter_collection.csv > profiles/r02_pmc_hbm_traffic_1m_laplace.json

The passes are separate runs (`rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py
--steps 3 --warmup 1 --no-cpu-baseline`, then the same with WRITE_SIZE).  rocprofv3 reports KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request of a wide (16 B per lane) coalesced load, so read bytes = 2 x FETCH_SIZE x 1024
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE x 1024 as is.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def collect(path):
    per = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            per[(short(row["Kernel_Name"]), row["Counter_Name"], int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    return per


def main(fetch_csv, write_csv):
    out = {"note": __doc__.strip().split("\n\n", 1)[1].replace("\n", " "), "kernels": {}}
    biggest = {}
    for path in (fetch_csv, write_csv):
        for (kernel, counter, grid), vals in collect(path).items():
            k = out["kernels"].setdefault(kernel, {}).setdefault(counter, {"launches": 0, "sum_KiB": 0.0, "max_KiB": 0.0})
            k["launches"] += len(vals)
            k["sum_KiB"] += sum(vals)
            k["max_KiB"] = max(k["max_KiB"], max(vals))
            # the product launches of one kernel differ by grid (phase A vs A2 of tile_gemv_tall): keep them apart
            biggest.setdefault((kernel, counter), {})[grid] = sum(vals) / len(vals)
    for kernel, counters in out["kernels"].items():
        for c in counters.values():
            c["avg_KiB"] = c.pop("sum_KiB") / c["launches"]

    def hbm_bytes(prefix):
        # largest-grid launch group of the kernel whose name starts with prefix (the product's main launch)
        tot = 0.0
        for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            cands = [(max(g.items(), key=lambda kv: kv[1])[1]) for (k, c), g in biggest.items() if k.startswith(prefix) and c == counter]
            if not cands:
                return None
            tot += factor * max(cands) * 1024.0
        return tot

    out["tile_gemv_wide_hbm_bytes_per_launch"] = hbm_bytes("hm::tile_gemv_wide<")
    out["tile_gemv_tall_phaseA_hbm_bytes_per_launch"] = hbm_bytes("hm::tile_gemv_tall_grouped<") or hbm_bytes("hm::tile_gemv_tall<")
    # the fingerprint of the kernel sources these passes were taken on: bench.py quotes the traffic only while it matches
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from bench import kernel_source_sha1

        out["kernel_source_sha1"] = kernel_source_sha1()
    except Exception:  # noqa: BLE001
        out["kernel_source_sha1"] = None
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
