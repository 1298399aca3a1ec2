#!/usr/bin/env python3
"""Per-layer-shape table of the forward's convolutions at the bench batch (64 x 512 x 512): calls per forward, chosen tile
configuration, its time, the MIOpen + epilogue time, TFLOP/s and the minimum-traffic GB/s.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("POSEPAF_TUNE_MIOPEN", "1")   # this table wants the MIOpen column too
import torch
from posepaf import fused_model as fm

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = fm.build_inference_model(torch.device("cuda", 0))
x = torch.rand(B, 512, 512, 3, device="cuda").half()
with torch.no_grad():
    model(x)
    fm._conv_calls.clear()
    model(x)
torch.cuda.synchronize()
rows = []
up2 = {k_: v for k_, v in fm._conv_timing.items() if k_[0] == "up2"}   # upsample -> 3x3 -> add(s): separate launches vs one (forward_up2)
for key, times in fm._conv_timing.items():
    if isinstance(key[0], str):   # fusion decisions ("up2", "dual", "pool", "mean", "cat"): listed below
        continue
    n, c, h, w, k, r, pad, dil, mode, act = key[:10]   # (+ ("slice", ldx, ldy) for the in-place halves of the concatenation)
    calls = fm._conv_calls.get(key, 0)
    best = fm._conv_choice[key]
    t = times["miopen"] if best < 0 else times[best]
    flop = 2.0 * n * h * w * c * k * r * r
    byts = 2.0 * n * h * w * (c + k * (2 if mode else 1)) + 2.0 * k * c * r * r
    rows.append((calls * t, key, calls, best, t, times.get("miopen", float("nan")), flop / t / 1e9, byts / t / 1e6))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{'c_in':>5} {'c_out':>5} {'hxw':>9} k pad dil mode act | calls cfg   best_ms  miopen_ms  TFLOP/s    GB/s  share")
for tt, key, calls, best, t, tm, tf, gb in rows:
    n, c, h, w, k, r, pad, dil, mode, act = key[:10]
    print(f"{c:5d} {k:5d} {h:4d}x{w:<4d} {r} {pad:3d} {dil:3d} {mode:4d} {int(act):3d} | {calls:5d} {best:3d} {t:9.3f} {tm:10.3f} {tf:8.0f} {gb:7.0f} {100 * tt / tot:5.1f}%")
print(f"sum over tuned shapes: {tot:.2f} ms per forward")
print("all configurations, ms (top 8 shapes):")
for tt, key, calls, best, t, tm, tf, gb in rows[:8]:
    print(key, {k: round(v, 3) for k, v in fm._conv_timing[key].items()})
print("1x1 shapes, every candidate (ms; 105 = the streaming 1x1 kernel, 101-103 = implicit GEMM, 0-9 = composable_kernel templates):")
for tt, key, calls, best, t, tm, tf, gb in rows:
    if key[5] == 1:
        tms = fm._conv_timing[key]
        ck = min([v for k_, v in tms.items() if isinstance(k_, int) and k_ < 100] or [float("nan")])
        print(f"  {key[1]:4d}->{key[4]:4d} @{key[2]}x{key[3]} mode {key[8]}: best ck {ck:.3f}  pw {tms.get(105, float('nan')):.3f}  igemm "
              f"{min([v for k_, v in tms.items() if isinstance(k_, int) and 101 <= k_ <= 103] or [float('nan')]):.3f}  -> {best}")
print("upsample x2 -> 3x3 convolution -> add(s): separate launches vs one launch of the halo kernel, ms")
for key, t in up2.items():
    _, n, c, h, w, k, two, act = key
    print(f"  {c:4d}->{k:4d} from {h}x{w} ({'two adds' if two else 'one add'}): " + "  ".join(f"{k_} {v:.3f}" for k_, v in t.items()) +
          f"  chosen {fm._conv_choice[key]} (0 separate, 1 halo kernel reads the upsample, 2.. collapsed 2x2 form, tile 256 / 128 / 64)")
print("convolution + second output (y, y + other): separate add vs one launch per own tile width, ms")
for key, t in fm._conv_timing.items():
    if key[0] == "dual":
        print(" ", key[1:], {k_: round(v, 3) for k_, v in t.items()}, "chosen", fm._conv_choice[key] or "separate")
print("last 1x1 + 1x1 skip of a residual block as one product over [t ; x]: separate vs fused, ms")
for key, t in fm._conv_timing.items():
    if key[0] == "cat":
        print(" ", key[1:], {k_: round(v, 3) for k_, v in t.items()}, "chosen", "fused" if fm._conv_choice[key] else "separate")
