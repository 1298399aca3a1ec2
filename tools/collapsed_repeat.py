#!/usr/bin/env python3
"""Race detector for pp_conv_up2_collapsed_f16 at the bench geometry: every variant launched repeatedly on the same operands, each
result BIT-compared with the first and the first compared with conv3x3(upsample2(x)) on a slice.  GPU only.
usage: collapsed_repeat.py [n 256] [repeats 12]      (POSEPAF_LIB selects an A/B build of the library)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from posepaf import _lib, fused_model as fm

L = _lib.load()
print("library", _lib.LIB_PATH)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
vp = C.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)
for ci, co, h, w in ((256, 256, 64, 64), (384, 384, 32, 32), (512, 512, 16, 16), (640, 640, 8, 8)):
    g = torch.Generator(device="cpu").manual_seed(61)
    conv = torch.nn.Conv2d(ci, co, 3, 1, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (ci * 9) ** 0.5)
        conv.bias.copy_(torch.randn(co, generator=g))
    f = fm.FConv(conv, None, True).cuda().half()
    x = torch.randn(N, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    e1 = torch.randn(N, co, 2 * h, 2 * w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    e2 = torch.randn(N, co, 2 * h, 2 * w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    w4 = f._collapsed_weights()
    up = F.interpolate(x[:4].float(), scale_factor=2, mode="nearest")
    act = F.leaky_relu(F.conv2d(up, f.weight.float(), f.bias.float(), 1, 1), 0.01)
    for mode in (2, 3):
        ref = act + e1[:4].float() if mode == 2 else act.half().float() + e1[:4].float() + e2[:4].float()
        for bn in (512, 256, 128):
            if co % (bn if bn != 512 else 64):
                continue
            first, bad, worst = None, 0, 0.0
            for r in range(R):
                y = torch.full((N, co, 2 * h, 2 * w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
                rc = L.pp_conv_up2_collapsed_f16(vp(x.data_ptr()), vp(w4.data_ptr()), vp(f.bias.data_ptr()), vp(e1.data_ptr()),
                                                 vp(e2.data_ptr()) if mode == 3 else None, vp(y.data_ptr()), N, h, w, ci, co, mode, 0.01, bn, st)
                if rc != 0:
                    break
                torch.cuda.synchronize()
                if first is None:
                    first = y
                    err = (y[:4].float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
                else:
                    d = (y.float() - first.float()).abs()
                    d = torch.nan_to_num(d, nan=1e9).max().item()
                    bad += d != 0.0
                    worst = max(worst, d)
            if rc != 0:
                print(f"{ci}->{co} from {h}x{w} mode {mode} bn {bn}: not taken ({rc})", flush=True)
                continue
            print(f"{ci}->{co} from {h}x{w} mode {mode} bn {bn}: first vs reference {err:.2e} of the scale; {bad} of {R - 1} repeats differ "
                  f"(max {worst:.4g})", flush=True)
