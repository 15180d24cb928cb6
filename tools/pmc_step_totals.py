#!/usr/bin/env python
"""Whole-run totals of rocprofv3 --pmc passes over bench.py (all kernels, all launches): MFMA-busy share of the SIMD cycles the
chip was active, and HBM-side bytes (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950 per
MI355X_MICROARCH.md).  Divided by the number of timed + warm-up steps of the run for per-step figures.

    python tools/pmc_step_totals.py <dir with mfma/ fetch/ write/ passes> <steps in the run>"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
root, steps = sys.argv[1], float(sys.argv[2])
tot = collections.Counter()
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot):
    print(f"{k:32s} {tot[k]:.4g}")
if tot["GRBM_GUI_ACTIVE"] > 0:
    act = tot["GRBM_GUI_ACTIVE"] / 8.0          # cycles per XCD (summed over the XCDs by the counter)
    print(f"MFMA-busy share of the active SIMD cycles (all kernels of the run): {tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 1024.0):.3f}")
    print(f"GPU-active time per step at 2.4 GHz: {act / 2.4e9 / steps * 1e3:.2f} ms")
if tot["FETCH_SIZE"] > 0 or tot["WRITE_SIZE"] > 0:
    rd, wr = tot["FETCH_SIZE"] * 1024 * 2 / steps, tot["WRITE_SIZE"] * 1024 / steps
    print(f"HBM-side traffic per step: read {rd / 1e9:.2f} GB (FETCH_SIZE x2), write {wr / 1e9:.2f} GB")
