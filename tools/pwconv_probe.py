#!/usr/bin/env python3
"""Micro-benchmark: fused MFMA 1x1 conv (pp_pwconv_f16) vs MIOpen conv + epilogue kernel; GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from posepaf import fused_model as fm

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (K, N, H, with_res) in [(256, 128, 128, False), (128, 256, 128, True), (64, 64, 256, False), (128, 128, 128, False),
                            (256, 192, 64, False), (192, 96, 64, False)]:
    conv = torch.nn.Conv2d(K, N, 1, bias=True)
    fc = fm.FConv(conv, None, True).cuda().half().to(memory_format=torch.channels_last)
    x = torch.randn(B, K, H, H, device="cuda").half().contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, N, H, H, device="cuda").half().contiguous(memory_format=torch.channels_last) if with_res else None
    fm.USE_PWCONV = True
    t_new = timeit(lambda: fc(x, res))
    y1 = fc(x, res)
    fm.USE_PWCONV = False
    t_old = timeit(lambda: fc(x, res))
    y0 = fc(x, res)
    byts = x.numel() * 2 + y1.numel() * 2 * (2 if with_res else 1)
    print(f"K={K:3d} N={N:3d} H={H:3d} res={with_res!s:5}: fused {t_new:8.1f} us ({byts / t_new / 1e6:6.2f} TB/s)   "
          f"miopen+epilogue {t_old:8.1f} us   speedup {t_old / t_new:4.2f}x   maxdiff {(y1.float() - y0.float()).abs().max().item():.4f}")
