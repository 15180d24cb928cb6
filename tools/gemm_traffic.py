#!/usr/bin/env python
"""HBM-side traffic per CALL of the encoder GEMM shapes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE — they do not fit one
pass) over tools/gemm_probe.py, written as the JSON bench.py's `roofline.traffic` reads.

    tools/run_pmc.sh gpurun_out/pmc_traffic_48000 tools/gemm_probe.py fetch write      (with LAKO_PROBE_TOKENS=48000)
    python tools/gemm_traffic.py gpurun_out/pmc_traffic_48000 48000 > profiles/r03_gemm_traffic.json

Units and corrections (MI355X_MICROARCH.md § HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads (16 B per lane: LDS-DMA and global loads alike) and is DOUBLED here; WRITE_SIZE is exact for 16-byte stores and float
atomics.  Infinity-Cache hits are counted, so these are fabric-side (L2-miss) bytes, an upper bound of the HBM bytes.  A call of the
library may be two dispatches (the 256x256 kernel on the rows of the full rounds + a small-tile launch on the row tail): dispatches
are grouped into calls by order — a call starts at a gemm_nt_kernel<…,2,4,{6,8,9},4,…> / gemm_tn256_kernel / gemm_nt4_kernel / gemm_tn4_kernel dispatch."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import CALLS, NT_SHAPES, TN_GROUP  # noqa: E402

csv.field_size_limit(1 << 30)


def dispatches(root, which):
    rows = []
    for path in glob.glob(os.path.join(root, which, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if "gemm_" in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
    per = collections.OrderedDict()
    for did, name, cn, val in sorted(rows):
        per.setdefault(did, [name, 0.0])[1] += val      # one row per XCD / dimension instance: summed
    return list(per.values())


def calls_of(disp):
    """[(kind, [kernel names], counter sum)] in dispatch order"""
    out = []
    for name, val in disp:
        head = ("gemm_nt_kernel" in name and re.search(r"Li2ELi4ELi[689]ELi4E", name) is not None) or "gemm_tn256_kernel" in name or \
            "gemm_nt4_kernel" in name or "gemm_tn4_kernel" in name   # 256- / 192- / 288-row tiles; round 6: the four-wave kernels (one dispatch per call)
        if head or not out:
            out.append(["tn" if "gemm_tn" in name else "nt", [name], val])
        else:
            out[-1][1].append(name)
            out[-1][2] += val
    return out


def main():
    root, rows = sys.argv[1], int(sys.argv[2])
    fetch, write = calls_of(dispatches(root, "fetch")), calls_of(dispatches(root, "write"))
    n_calls = CALLS * (len(NT_SHAPES) + 1)
    if len(fetch) != n_calls or len(write) != n_calls:
        raise SystemExit(f"expected {n_calls} calls, found {len(fetch)} (fetch) / {len(write)} (write): "
                         f"{[c[1] for c in fetch]}")
    shapes, wsum, tsum, asum = [], 0, 0.0, 0.0
    for i, (N, K, per_layer) in enumerate(NT_SHAPES):
        fc, wc = fetch[i * CALLS:(i + 1) * CALLS], write[i * CALLS:(i + 1) * CALLS]
        fetch_raw = sum(c[2] for c in fc) / CALLS * 1024
        wr = sum(c[2] for c in wc) / CALLS * 1024
        alg_r, alg_w = (rows * K + N * K) * 2, rows * N * 2
        traffic = 2 * fetch_raw + wr
        shapes.append({"kernel": "gemm_nt: " + " + ".join(sorted({n.split("(")[0][-60:] for n in fc[0][1]})), "dispatches_per_call": len(fc[0][1]),
                       "M": rows, "N": N, "K": K, "launches_per_encoder_layer_and_step": per_layer,
                       "fetch_size_raw_bytes": round(fetch_raw), "fetch_bytes_gfx950_corrected": round(2 * fetch_raw), "write_bytes": round(wr),
                       "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w, "traffic_bytes": round(traffic),
                       "algorithmic_bytes": alg_r + alg_w, "traffic_over_algorithmic": round(traffic / (alg_r + alg_w), 3)})
        wsum += per_layer
        tsum += per_layer * traffic
        asum += per_layer * (alg_r + alg_w)
    fc, wc = fetch[len(NT_SHAPES) * CALLS:], write[len(NT_SHAPES) * CALLS:]
    fetch_raw = sum(c[2] for c in fc) / CALLS * 1024
    wr = sum(c[2] for c in wc) / CALLS * 1024
    alg_r = sum(rows * (M + N) * 2 for M, N in TN_GROUP)
    alg_w = sum(M * N * 4 for M, N in TN_GROUP)
    tn = {"kernel": "gemm_tn4_kernel / gemm_tn256_kernel (grouped launch: the four weight gradients of an encoder layer, K = rows)", "K": rows,
          "items_MxN": TN_GROUP, "fetch_size_raw_bytes": round(fetch_raw), "fetch_bytes_gfx950_corrected": round(2 * fetch_raw),
          "write_bytes": round(wr), "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w,
          "traffic_bytes": round(2 * fetch_raw + wr), "algorithmic_bytes": alg_r + alg_w,
          "traffic_over_algorithmic": round((2 * fetch_raw + wr) / (alg_r + alg_w), 3),
          "write_amplification": round(wr / alg_w, 2),
          "write_note": "split-K partial tiles are added into the fp32 gradient by float atomics: one 256 KiB pass over each output tile per K-split"}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/gemm_probe.py, per call of the library (mean of "
                       f"{CALLS}); FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md § HBM); fabric-side bytes incl. Infinity-Cache hits; "
                       "nt_mean_*: mean over the encoder's NT shapes weighted by their launches per layer and step",
               "rows": rows, "shapes": shapes, "tn_grouped": tn,
               "nt_mean_traffic_bytes_per_launch": round(tsum / wsum), "nt_mean_algorithmic_bytes_per_launch": round(asum / wsum),
               "nt_traffic_over_algorithmic": round(tsum / asum, 3)}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
