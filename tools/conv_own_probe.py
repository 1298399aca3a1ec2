#!/usr/bin/env python3
"""Own implicit-GEMM convolution (pp_conv_own_f16) against the composable_kernel template configurations (pp_conv_f16) and
MIOpen + epilogue on the forward's hottest layer shapes at the bench batch; GPU only."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [  # c_in, c_out, h, w, k, pad, dil, mode
    (256, 256, 128, 128, 3, 1, 1, 0), (256, 256, 128, 128, 3, 1, 1, 2), (384, 384, 64, 64, 3, 1, 1, 2), (192, 192, 64, 64, 3, 1, 1, 0),
    (128, 128, 128, 128, 3, 1, 1, 0), (512, 512, 32, 32, 3, 1, 1, 2), (384, 256, 64, 64, 3, 1, 1, 0), (256, 256, 128, 128, 1, 0, 1, 1),
    (128, 256, 128, 128, 1, 0, 1, 1), (256, 128, 128, 128, 1, 0, 1, 0), (64, 64, 256, 256, 3, 1, 1, 0),
]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for (ci, co, h, w, k, pad, dil, mode) in SHAPES:
    x = torch.randn(N, ci, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda").half()
    ex = torch.randn(N, co, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last) if mode else None
    y = torch.empty((N, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    st = vp(torch.cuda.current_stream().cuda_stream)
    flop = 2.0 * N * h * w * ci * co * k * k
    res = {}
    for cfg in range(L.pp_conv_num_configs()):
        args = (vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None, vp(y.data_ptr()), N, h, w,
                ci, co, k, pad, dil, mode, 0.01, cfg, st)
        if L.pp_conv_f16(*args) != 0:
            continue
        res[f"ck{cfg}"] = timed(lambda: L.pp_conv_f16(*args))
    yck = y.clone()
    for bn in (512, 256, 128, 64):
        if bn < 512 and co % bn:
            continue
        args = (vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None, vp(y.data_ptr()), N, h, w,
                ci, co, k, pad, dil, mode, 0.01, bn, st)
        if L.pp_conv_own_f16(*args) != 0:
            continue
        res[f"own{bn}"] = timed(lambda: L.pp_conv_own_f16(*args))
        torch.cuda.synchronize()
        err = (y.float() - yck.float()).abs().max().item()
        res[f"own{bn}_maxdiff_vs_ck"] = err
    best_ck = min(v for kk, v in res.items() if kk.startswith("ck"))
    best_own = min(v for kk, v in res.items() if kk.startswith("own") and "diff" not in kk)
    print(f"{ci:4d}->{co:4d} {h:3d}x{w:<3d} k{k} d{dil} mode{mode}: best CK {best_ck:7.3f} ms ({flop / best_ck / 1e9:6.0f} TF)  "
          f"own {best_own:7.3f} ms ({flop / best_own / 1e9:6.0f} TF)  ratio {best_ck / best_own:5.2f} | " +
          " ".join(f"{kk}={v:.3f}" for kk, v in res.items()), flush=True)
