#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of K_A / K_B / K_C (diagnostic stamps); GPU only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from posepaf import _lib, synth
from posepaf.api import PosePostProcessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = _lib.load()
post = PosePostProcessor(max_batch=B, max_h=128, max_w=128, max_peaks_per_part=64)
for P in (6, 15, 30):
    nets = np.stack([synth.make_net_output(P, 500 + i, dtype=np.float16) for i in range(16)])
    dev = torch.from_numpy(np.concatenate([nets] * (B // 16))).cuda()
    post.process(dev, 512)
    for name, nwg, labels in (("K_A", 18 * B, ["load", "nms", "refine"]), ("K_B", 30 * B, ["load", "score", "rank/sort", "greedy", "output"]),
                               ("K_C", B, ["load", "limbs", "records"])):
        st = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
        L.pp_debug_set_stamps(C.c_void_p(st.data_ptr()))
        if name == "K_A":
            post.nms(dev)
        else:
            post.process(dev, 512)
        torch.cuda.synchronize()
        L.pp_debug_set_stamps(None)
        s = st.cpu().numpy().reshape(nwg, 8)
        ok = (s[:, :len(labels) + 1] > 0).all(axis=1)   # workgroups that ran every phase (empty limbs return early)
        s = s[ok]
        raw = s
        d = np.diff(s[:, :len(labels) + 1], axis=1)
        span = (s[:, len(labels)].max() - s[:, 0].min())
        if name == "K_C":
            hit = np.nonzero(raw[:16, 6])[0]
            if len(hit):
                print(f"P={P:2d} K_C: id-summing merges in scenes (P, seed): " + ", ".join(f"({P},{500 + int(i)})" for i in hit))
            print(f"P={P:2d} K_C: one-by-one connections/img mean {raw[:, 4].mean():5.1f} max {raw[:, 4].max():3d}; cycles in them mean {raw[:, 5].mean()/1e3:6.1f} kcyc; id-summing merges (tables abandoned) total {int(raw[:, 6].sum())}")
        if name == "K_B":
            tot = s[:, len(labels)] - s[:, 0]
            for w in np.argsort(-tot)[:4]:
                print(f"P={P:2d} K_B slowest wg: total {tot[w]/1e3:6.1f} kcyc = " + "  ".join(f"{lab} {d[w, i]/1e3:6.1f}" for i, lab in enumerate(labels)))
        print(f"P={P:2d} {name}: wgs {ok.sum():5d}  kernel span {span/1e3:8.1f} kcyc | " +
              "  ".join(f"{lab} mean {d[:, i].mean()/1e3:6.1f} max {d[:, i].max()/1e3:6.1f}" for i, lab in enumerate(labels)) +
              f" | wg total mean {(s[:, len(labels)] - s[:, 0]).mean()/1e3:6.1f} max {(s[:, len(labels)] - s[:, 0]).max()/1e3:6.1f} kcyc")
