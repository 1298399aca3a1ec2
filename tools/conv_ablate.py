#!/usr/bin/env python3
"""Ablation of the halo-tile 3x3 kernel on the hot layer (256 -> 256 @ 128 x 128, batch 128): time with parts switched off
(POSEPAF_CONV_DBG = 1 no DMA in the loop, 4 no fragment reads, 5 MFMA + barriers only, 6 DMA + barriers only, 7 barriers only, 15 ... and no epilogue stores, 64 / 71 full-line store pattern, 1024 (+4, +5, +128) in-kernel section stamps and clock).  Run once per setting (the env is read once)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
from posepaf import _lib
L = _lib.load()
vp = C.c_void_p
N, ci, co, h, w = 128, int(os.environ.get('CI', '256')), 256, 128, 128
x = torch.randn(N, ci, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
wt = (torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5).half().contiguous(memory_format=torch.channels_last)
b = torch.randn(co, device="cuda").half()
y = torch.empty((N, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
st = vp(torch.cuda.current_stream().cuda_stream)
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 512
args = (vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), N, h, w, ci, co, 3, 1, 1, 0, 0.01, bn, st)
assert L.pp_conv_own_f16(*args) == 0
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.pp_conv_own_f16(*args); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts)[3]
if int(os.environ.get('POSEPAF_CONV_DBG', '0')) & 1024:
    for _ in range(300):   # clocks settle under sustained load
        L.pp_conv_own_f16(*args)
    torch.cuda.synchronize()
    out = (C.c_double * 6)()
    L.pp_conv_debug_clock.argtypes = [C.POINTER(C.c_double), C.c_int]
    assert L.pp_conv_debug_clock(out, 256) == 0
    ghz = out[2] / out[5] / 10
    tot = sum(out[:5])
    names = ["wait first DMA + stores", "first fragment reads", "main loop", "next decode + DMA issue", "epilogue issue"]
    print(f"  in-kernel clock {ghz:.3f} GHz (main loop); per workgroup, all its tiles: {tot / ghz / 1e3:.1f} us")
    for nm, v in zip(names, out[:5]):
        print(f"    {nm:26s} {v / ghz / 1e3:8.1f} us  {100 * v / tot:5.1f} %")
print(f"ci={ci} dbg={os.environ.get('POSEPAF_CONV_DBG', '0'):>2} bn={bn}: {t:.3f} ms  ({2.0 * N * h * w * ci * co * 9 / t / 1e9:.0f} TF-equivalent)")
