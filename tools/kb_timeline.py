#!/usr/bin/env python3
"""Timeline of the fused limb kernel K_B (k_limb_connect + the per-image assembly tail) on bench.py's scene mix: when each
image's limb workgroups start / end and when its assembly starts / ends, from in-kernel stamps of the chip-wide 100 MHz
counter (POSEPAF_STAMP_REALTIME=1).  GPU only.   usage: kb_timeline.py [B] [threads 256|512]"""
import ctypes as C
import json
import os
import sys

os.environ["POSEPAF_STAMP_REALTIME"] = "1"
if len(sys.argv) > 2:
    os.environ["POSEPAF_KB_THREADS"] = sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
from posepaf import _lib
from posepaf.api import PosePostProcessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = _lib.load()
post = PosePostProcessor(max_batch=B, max_h=128, max_w=128, max_peaks_per_part=64)
mix, _ = bench.build_scenes(B)
dev = torch.from_numpy(mix).cuda()
for _ in range(3):
    post.process(dev, 512)
ms = post.time_kernels(dev, 512, True, iters=30)
print("events:", {k: round(v * 1e3, 1) for k, v in ms.items()}, "us")
nwg = 31 * B          # fused form: grid (31, B) -- workgroup 30 of an image is its assembly
st = torch.zeros((nwg + B) * 8, dtype=torch.int64, device="cuda")
L.pp_debug_set_stamps(C.c_void_p(st.data_ptr()))
post.process(dev, 512)
torch.cuda.synchronize()
L.pp_debug_set_stamps(None)
s = st.cpu().numpy().reshape(nwg + B, 8).astype(np.float64)
limb, tail = s[:nwg].reshape(B, 31, 8)[:, :30], s[nwg:]          # limb[position, limb]: workgroup (limb, position) -> image order[position]
t0 = limb[:, :, 0][limb[:, :, 0] > 0].min()
us = lambda v: (v - t0) / 100.0
ran = limb[:, :, 5] > 0                                   # workgroups that scored pairs (empty limbs return early, no end stamp)
end = np.where(ran, limb[:, :, 5], limb[:, :, 0])
print(f"B={B} threads={os.environ.get('POSEPAF_KB_THREADS', '256')}: kernel span by stamps {us(max(end.max(), tail[:, 3].max())):.1f} us; "
      f"last limb end {us(end.max()):.1f} us; last assembly end {us(tail[:, 3].max()):.1f} us")
# per position (dispatch order = heaviest first): limb start span, limb end, assembly start/end
tail_by_pos = None
print("pos  limb_start(min..max)   limb_end(max)  slowest_limb_us(load/score/sort/greedy/out)")
for pos in list(range(0, 12)) + list(range(12, B, max(1, B // 12))):
    r = ran[pos]
    if not r.any():
        continue
    st0, e = limb[pos, :, 0], end[pos]
    w = int(np.argmax(np.where(r, limb[pos, :, 5] - limb[pos, :, 0], 0)))
    d = np.diff(limb[pos, w, :6]) / 100.0
    print(f"{pos:3d}  {us(st0.min()):7.1f} .. {us(st0.max()):7.1f}   {us(e.max()):7.1f}        limb {w:2d}: " + " / ".join(f"{x:5.1f}" for x in d))
a0, a3 = tail[:, 0], tail[:, 3]
ok = a0 > 0
print(f"assembly (by image): start min {us(a0[ok].min()):.1f} max {us(a0[ok].max()):.1f}; duration mean {((a3 - a0)[ok] / 100).mean():.1f} "
      f"max {((a3 - a0)[ok] / 100).max():.1f} us; end max {us(a3[ok].max()):.1f}")
longest = np.argsort(-(a3 - a0))[:8]
for i in longest:
    print(f"  image {int(i):3d}: assembly {us(a0[i]):7.1f} -> {us(a3[i]):7.1f} us  (init {(tail[i,1]-tail[i,0])/100:5.1f}, limbs {(tail[i,2]-tail[i,1])/100:5.1f}, records {(tail[i,3]-tail[i,2])/100:5.1f})")
# occupancy over time: limb workgroups alive per 10-us bin
bins = np.arange(0, us(max(end.max(), a3.max())) + 10, 10)
alive = [(int(((us(limb[:, :, 0]) <= t) & (us(end) > t) & ran).sum()), int(((us(a0) <= t) & (us(a3) > t) & ok).sum())) for t in bins]
print("t_us: limb WGs alive / assemblies alive")
print("  ".join(f"{int(t)}:{a}/{b}" for t, (a, b) in zip(bins, alive)))
