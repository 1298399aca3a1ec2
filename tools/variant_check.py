#!/usr/bin/env python3
"""Every kernel variant the tuner may pick for a layer shape, at the bench batch, against an fp32 torch convolution of the same
fp16 operands.  usage: variant_check.py "n,c,h,w,k,r,pad,dil,mode" ...   GPU only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from posepaf import _lib

L = _lib.load()
vp = C.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (256, 192, 32, 32, 384, 1, 0, 1, 1), (256, 640, 16, 16, 320, 1, 0, 1, 0), (256, 768, 8, 8, 320, 1, 0, 1, 0),
    (256, 64, 256, 256, 128, 1, 0, 1, 1)]
bad = 0
for n, c, h, w, k, r, pad, dil, mode in shapes:
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(n, c, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(k, c, r, r, generator=g) / (c * r * r) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(k, generator=g).cuda().half()
    ex = torch.randn(n, k, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    ref = None
    for lo in range(0, n, 32):   # fp32 reference in slices (memory)
        part = F.conv2d(x[lo:lo + 32].float(), wt.float(), b.float(), 1, pad, dil)
        if mode == 1:
            part = part + ex[lo:lo + 32].float()
        part = F.leaky_relu(part, 0.01)
        if mode == 2:
            part = part + ex[lo:lo + 32].float()
        ref = part.half() if ref is None else torch.cat([ref, part.half()])
    scale = max(1.0, ref.float().abs().max().item())
    for cfg in list(range(L.pp_conv_num_configs())) + [101, 102, 103, 104, 105]:
        y = torch.full((n, k, h, w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
        e = vp(ex.data_ptr()) if mode else None
        if cfg == 105:
            if r != 1:
                continue
            rc = L.pp_pw_f16(vp(x.data_ptr()), None, vp(wt.data_ptr()), vp(b.data_ptr()), e, None, vp(y.data_ptr()), None, n * h * w, h * w,
                             c, k, k, mode, 0.01, st)
        elif cfg >= 100:
            rc = L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), e, vp(y.data_ptr()), n, h, w, c, k, r, pad, dil, mode,
                                   0.01, {101: 256, 102: 128, 103: 64, 104: 512}[cfg], st)
        else:
            rc = L.pp_conv_ld_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), e, vp(y.data_ptr()), n, h, w, c, k, r, pad, dil, mode,
                                  0.01, cfg, c, k, st)
        if rc != 0:
            continue
        torch.cuda.synchronize()
        err = 0.0
        for lo in range(0, n, 32):
            d = (y[lo:lo + 32].float() - ref[lo:lo + 32].float()).abs()
            err = max(err, float(d.max()) if torch.isfinite(d).all() else float("inf"))
        flag = "" if err <= 3e-3 * scale else "   <-- WRONG"
        bad += bool(flag)
        print(f"{(n, c, h, w, k, r, pad, dil, mode)} cfg {cfg}: max err {err:.4g} (scale {scale:.3g}){flag}", flush=True)
print("wrong variants:", bad)
