#!/usr/bin/env python3
"""Post-processing chain timing (HIP events via pp_time_kernels) for the three launch structures of pp_process_batch
(pp_debug_set_mode) on bench.py's scene mix and on uniform densities; GPU only.  Also checks that every mode returns
byte-identical records."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
from posepaf import synth
from posepaf.api import PosePostProcessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
post = PosePostProcessor(max_batch=B, max_h=128, max_w=128, max_peaks_per_part=64)
out = {}
mix, _ = bench.build_scenes(B)
cases = {"bench mix": torch.from_numpy(mix).cuda()}
for P in (2, 6, 15, 30):
    nets = np.stack([synth.make_net_output(P, 500 + i, dtype=np.float16) for i in range(16)])
    cases[f"{P} people"] = torch.from_numpy(np.concatenate([nets] * (B // 16))).cuda()
names = {0: "fused+ordered", 1: "separate assembly launch", 2: "fused, natural order"}
for cname, dev in cases.items():
    ref = None
    for mode in (0, 1, 2):
        post.set_mode(mode)
        rec = post.process(dev, 512).copy()
        if ref is None:
            ref = rec
        same = all(rec[i]["n_humans"] == ref[i]["n_humans"] and rec[i]["status"] == ref[i]["status"] and
                   rec[i]["humans"][:rec[i]["n_humans"]].tobytes() == ref[i]["humans"][:ref[i]["n_humans"]].tobytes() for i in range(B))
        ms = post.time_kernels(dev, 512, True, iters=30)
        out[f"{cname} / {names[mode]}"] = {k: round(v * 1e3, 1) for k, v in ms.items()} | {"identical_records": bool(same)}
        print(f"{cname:10s} {names[mode]:26s} " + "  ".join(f"{k} {v * 1e3:7.1f} us" for k, v in ms.items()) + f"  same={same}", flush=True)
post.set_mode(0)
print(json.dumps(out))
