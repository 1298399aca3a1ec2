#!/usr/bin/env python3
"""Merge two rocprofv3 --pmc passes (FETCH_SIZE alone, WRITE_SIZE alone) of `bench.py --postproc-only --batch B` into
profiles/r02_pmc_traffic.json: HBM bytes per launch of K_A (k_heat_peaks) and K_B (k_limb_connect, which includes the person
assembly since round 2).

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <B> [output name, default r03_pmc_traffic.json]

rocprofv3 reports these counters in KiB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced streaming reads
(MI355X_MICROARCH.md, section HBM), so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import glob
import json
import os
import sys

fdir, wdir, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"k_heat_peaks": "k_heat_peaks", "k_limb_connect": "k_limb_connect<"}


def per_launch(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> dispatch id -> value (summed over XCDs)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            for name, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    acc[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}


fetch, write = per_launch(fdir, "FETCH_SIZE"), per_launch(wdir, "WRITE_SIZE")
out_path = os.path.join(ROOT, "profiles", sys.argv[4] if len(sys.argv) > 4 else "r03_pmc_traffic.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {
    "_note": "HBM traffic per launch from rocprofv3 --pmc (separate passes: FETCH_SIZE alone, WRITE_SIZE alone) of `bench.py "
             "--postproc-only --batch B`, merged by tools/pmc_traffic.py. rocprofv3 reports KiB. On gfx950 FETCH_SIZE counts exactly "
             "half of the bytes of a wide coalesced (16 B/lane) streaming read (MI355X_MICROARCH.md, section HBM), so traffic = "
             "(2*FETCH_SIZE + WRITE_SIZE)*1024."}
out.setdefault("_raw_kib_by_batch", {})[f"batch{B}"] = {k: {"FETCH_SIZE": fetch[k], "WRITE_SIZE": write[k]} for k in fetch}
for k in fetch:
    out.setdefault(k, {})[f"batch{B}"] = (2 * fetch[k] + write[k]) * 1024
out.setdefault("_algorithmic_bytes", {})[f"batch{B}"] = {"k_heat_peaks": B * 18 * 2 * 128 * 128 * 2, "k_limb_connect": B * 30 * 2 * 128 * 128 * 2}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: out[k][f"batch{B}"] for k in fetch}))
