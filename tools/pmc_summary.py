#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters (one pass = one directory), summed over the XCDs / instances of a dispatch and
averaged over the dispatches of each kernel whose name contains one of the given patterns.

    python tools/pmc_summary.py <out.json> <pattern[,pattern...] or pattern;pattern...> <pass dir> [<pass dir> ...]
"""
import collections
import csv
import glob
import json
import os
import sys

out_path, dirs = sys.argv[1], sys.argv[3:]
pats = sys.argv[2].split(";") if ";" in sys.argv[2] else sys.argv[2].split(",")   # (template names hold commas: use ";" for them)
res = {p: {} for p in pats}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))   # (pattern, counter) -> dispatch -> sum
        with open(f) as fh:
            for r in csv.DictReader(fh):
                for p in pats:
                    if p in r["Kernel_Name"]:
                        acc[(p, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (p, c), v in acc.items():
            vals = sorted(v.values())
            res[p][c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
for p, cs in res.items():
    g = lambda n: cs.get(n, {}).get("mean_per_launch")
    der = {}
    if g("SQ_BUSY_CU_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        # MFMA busy cycles are counted per SIMD, CU busy cycles per CU (4 SIMDs)
        der["mfma_pipe_busy_fraction"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES"))
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
        der["wave_cycles_waiting_fraction"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        der["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    if g("SQ_INSTS_LDS") and g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_cycles_over_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if der:
        cs["_derived"] = der
json.dump(res, open(out_path, "w"), indent=1)
for p, cs in res.items():
    print(p)
    for c, v in sorted(cs.items()):
        print("   ", c, v if c == "_derived" else round(v["mean_per_launch"], 1))
