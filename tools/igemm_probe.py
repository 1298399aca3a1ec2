"""Timings of the own implicit-GEMM kernel (pp_conv_own_f16, bn = 256 / 128 / 64) on the 1x1, dilated and 192-channel layer shapes that
stay on the composable_kernel template in the tuned model.  GPU only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
from posepaf import _lib
L = _lib.load()
vp = C.c_void_p
def run(N, ci, co, h, w, k, pad, dil, mode, bn):
    x = torch.randn(N, ci, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda").half()
    e = torch.randn(N, co, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(e)
    st = vp(torch.cuda.current_stream().cuda_stream)
    args = (vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(e.data_ptr()) if mode else None, vp(y.data_ptr()), N, h, w, ci, co, k, pad, dil, mode, 0.01, bn, st)
    assert L.pp_conv_own_f16(*args) == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); L.pp_conv_own_f16(*args); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[4]
    byts = 2.0 * N * h * w * (ci + co * (2 if mode else 1))
    print(f"{ci}->{co} {h}x{w} k{k} d{dil} mode{mode} bn{bn} N{N}: {t:.3f} ms  {byts / t / 1e6:.0f} GB/s  {2.0 * N * h * w * ci * co * k * k / t / 1e9:.0f} TF")
for bn in (256, 128):
    run(64, 128, 256, 128, 128, 1, 0, 1, 1, bn)
    run(64, 256, 256, 128, 128, 1, 0, 1, 1, bn)
    run(64, 256, 128, 128, 128, 1, 0, 1, 0, 128)
run(64, 192, 384, 64, 64, 1, 0, 1, 1, 128)
run(64, 128, 128, 128, 128, 3, 5, 5, 0, 128)
run(64, 192, 192, 64, 64, 3, 1, 1, 0, 64)
