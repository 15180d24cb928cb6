#!/usr/bin/env python
"""Mean per launch of every collected counter, per kernel, over the rocprofv3 --pmc passes under a directory
(each pass: <dir>/<pass>/…_counter_collection.csv, written with --output-format csv), plus a few derived ratios.
Only this library's kernels (names containing one of --match) are kept; template arguments are shortened.

    python tools/pmc_summary.py gpurun_out/pmc_r02_attn --match attn_ > profiles/r02_attn_pmc.txt
SQ_* cycle counters are in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (cycles) — MI355X_MICROARCH.md."""
import argparse
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void\s+)?([\w:]+)(<.*>)?\(", name)
    if m:
        targs = (m.group(2) or "")
        targs = targs.replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16")
        return m.group(1) + (targs if len(targs) < 60 else targs[:57] + "...>")
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)(\w+)", name)      # mangled: take the kernel identifier + template digits
    if m:
        n = int(m.group(1))
        ident, rest = m.group(2)[:n], m.group(2)[n:]
        return ident + "<" + re.sub(r"DF16b", "bf16,", re.sub(r"^I|E+v.*$", "", rest)).replace("Li", "").replace("E", ",").replace("Lb", "b") + ">"
    return name[:80]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--match", default="attn_,gemm_")
    args = ap.parse_args()
    keys = args.match.split(",")
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    meta = {}
    for path in sorted(glob.glob(os.path.join(args.root, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                kn = row["Kernel_Name"]
                if not any(k in kn for k in keys):
                    continue
                s = short(kn)
                a = acc[s][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                meta[s] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Workgroup_Size"])
    ratios = [("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"), ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
              ("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"), ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"), ("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"),
              ("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"), ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")]
    for s in sorted(acc):
        v, av, sg, lds, wg = meta[s]
        print(f"{s}\n  (VGPR {v} AGPR {av} SGPR {sg} LDS {lds} B, workgroup {wg})")
        mean = {c: t / n for c, (t, n) in acc[s].items()}
        for c in sorted(mean):
            print(f"  {c:36s} {mean[c]:16.0f}   (launches {acc[s][c][1]})")
        for a, b in ratios:
            if a in mean and b in mean and mean[b] > 0:
                print(f"     {a}/{b} = {mean[a] / mean[b]:.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA-busy cycles are summed over the 1024 SIMDs of the chip
            util = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            print(f"     MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 · 1024 SIMDs) = {util:.3f}")
        print()


if __name__ == "__main__":
    sys.exit(main())
