#!/usr/bin/env python3
"""Race detector for the fused forward: the same batch through the tuned model N times; every output must be BIT-identical to the
first (the kernels are deterministic: no atomics on the data path).  With `bisect`: the same per rewrite / fusion switch turned
off.  usage: forward_repeat.py [images 128] [repeats 10] [bisect]   GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
from posepaf import fused_model as fm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(3)
x = torch.rand(2 * B, 512, 512, 3, generator=g).half().to(dev)


def run(tag):
    model = fm.build_inference_model(dev)
    with torch.no_grad():
        for _ in range(2):
            first = model(x).clone()
        bad, worst = 0, 0.0
        for i in range(R):
            y = model(x)
            d = (y.float() - first.float()).abs().max().item()
            bad += d != 0.0
            worst = max(worst, d)
        print(f"{tag}: {bad} of {R} repeats differ from the first (max |diff| {worst:.4g}, output max {first.float().abs().max().item():.4g})",
              flush=True)
    del model
    return bad


run("all switches on")
if len(sys.argv) > 3:
    for flag in ("USE_COLLAPSED_UP2", "USE_FOLDED_MERGE", "USE_CAT_SKIP", "USE_POOL_FUSION", "USE_SUM_FUSION", "USE_SLICE_OUTPUT", "USE_PW",
                 "USE_OWN_CONV"):
        setattr(fm, flag, False)
        run(flag + " = False (and those above)")
