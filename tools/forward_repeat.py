#!/usr/bin/env python3
"""Race detector for the fused forward: the same batch through the tuned model N times; every output must be BIT-identical to the
first (the kernels are deterministic: no atomics on the data path).  With `bisect`: the same per rewrite / fusion switch turned
off.  (Inputs whose coarsest maps fall below 8 x 8 -- 256 x 256 images -- leave a few layer shapes to MIOpen, some of whose
fp16 kernels accumulate with atomics: those runs differ by an ulp or two per repeat even with every kernel of this library off.)
usage: forward_repeat.py [images 128] [repeats 10] [bisect|-] [side 512]   GPU only."""
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
S = int(sys.argv[4]) if len(sys.argv) > 4 else 512
x = torch.rand(2 * B, S, S, 3, generator=g).half().to(dev)


def run(tag):
    model = fm.build_inference_model(dev)
    with torch.no_grad():
        for _ in range(2):
            first = model(x).clone()
        bad, worst, chain, prev = 0, 0.0, 0, None
        for i in range(R):
            y = model(x).clone()
            d = (y.float() - first.float()).abs().max().item()
            bad += d != 0.0
            worst = max(worst, d)
            if prev is not None:
                chain += not torch.equal(y, prev)
            prev = y
        print(f"{tag}: {bad} of {R} repeats differ from the first (max |diff| {worst:.4g}, output max {first.float().abs().max().item():.4g}); "
              f"{chain} of {R - 1} differ from the repeat before them", flush=True)
    del model
    return bad


run("all switches on")
if len(sys.argv) > 3 and sys.argv[3] == "bisect":
    for flag in ("USE_COLLAPSED_UP2", "USE_FOLDED_MERGE", "USE_CAT_SKIP", "USE_POOL_FUSION", "USE_SUM_FUSION", "USE_SLICE_OUTPUT", "USE_PW",
                 "USE_OWN_CONV"):
        setattr(fm, flag, False)
        run(flag + " = False (and those above)")
