#!/usr/bin/env python3
"""Kernel timing probe (HIP events via pp_time_kernels) versus scene density; GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from posepaf import synth
from posepaf.api import PosePostProcessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
post = PosePostProcessor(max_batch=B, max_h=128, max_w=128, max_peaks_per_part=64)
for dtype in (np.float16, np.float32):
    for P in (0, 2, 6, 10, 15, 30):
        nets = np.stack([synth.make_net_output(P, 500 + i, dtype=dtype) for i in range(min(B, 16))])
        nets = np.concatenate([nets] * (B // len(nets)))
        dev = torch.from_numpy(nets).cuda()
        ms = post.time_kernels(dev, 512, True, iters=20)
        rec = post.process(dev, 512)
        nbytes = nets.nbytes * 48 / 50
        print(f"{np.dtype(dtype).name} P={P:2d} B={B} peaks/img={rec['n_peaks'].mean():6.1f} conns/img={rec['n_connections'].mean():6.1f} "
              f"humans/img={rec['n_humans'].mean():5.1f}  K_A {ms['k_heat_peaks']*1e3:7.1f}us  K_B {ms['k_limb_connect']*1e3:7.1f}us  "
              f"K_C(alone) {ms['k_assemble_wave']*1e3:7.1f}us chain {ms['chain']*1e3:7.1f}us  | K_A+K_B eff {nbytes/((ms['k_heat_peaks']+ms['k_limb_connect'])*1e-3)/1e9:7.1f} GB/s "
              f"status={int(np.bitwise_or.reduce(rec['status']))}", flush=True)
