"""Static instruction counts per kernel of csrc/device.hip compiled for gfx950 -> profiles/r02_isa_instruction_counts.txt

    python tools/isa_counts.py > profiles/r02_isa_instruction_counts.txt

Compiles the device translation unit to assembly with the flags of the build (no GPU needed) and counts, per kernel, the
mnemonics that tell how a kernel is built: matrix-core MFMAs, 16-byte global loads, fp64 FMAs, v_readlane broadcasts,
ds_bpermute exchanges, barriers, DPP moves, scratch accesses; plus the register counts of the kernel descriptor."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "htool_python_amd", "csrc", "device.hip")
COLS = [("v_mfma_f64_16x16x4", r"v_mfma_f64_16x16x4"), ("global_load_dwordx4", r"global_load_dwordx4"), ("v_fma_f64", r"v_fma(c)?_f64"),
        ("v_readlane_b32", r"v_readlane_b32"), ("ds_bpermute_b32", r"ds_bpermute_b32"), ("s_barrier", r"s_barrier"), ("_dpp", r"_dpp"), ("scratch_", r"scratch_")]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"^void ", "", o).split("(")[0] for o in out]


def main():
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "device.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-S", "--cuda-device-only", SRC, "-o", asm]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    bodies = {}
    for m in re.finditer(r"^(_ZN2hm\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    regs = {}
    meta = text[text.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in text else ""
    for blk in re.split(r"\n  - (?=\.)", meta):  # one block per kernel descriptor (keys in alphabetical order)
        m = re.search(r"\.name:\s+(_ZN2hm\w+)", blk)
        if not m:
            continue
        get = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        regs[m.group(1)] = (get("vgpr_count"), get("agpr_count"), get("vgpr_spill_count"), get("group_segment_fixed_size"))
    names = sorted(bodies)
    pretty = demangle(names)
    print("# static instruction counts per kernel of csrc/device.hip compiled for gfx950, made by tools/isa_counts.py")
    print("# (" + " ".join(cmd[:-3]) + " csrc/device.hip)")
    print("# columns: " + " | ".join(c for c, _ in COLS) + " || vgpr (arch + acc) | agpr | spilled vgprs | LDS bytes")
    for n, p in sorted(zip(names, pretty), key=lambda t: t[1]):
        counts = [len(re.findall(r"^\s+\S*" + rx, bodies[n], re.M)) for _, rx in COLS]
        r = regs.get(n, ("?",) * 4)
        print("%-72s %s || %s" % (p, " | ".join(str(c) for c in counts), " | ".join(r)))


if __name__ == "__main__":
    sys.exit(main())
