#!/usr/bin/env python3
"""Shader-clock breakdown of the person assembly (k_assemble_wave as its own launch, one wave per image); GPU only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from posepaf import _lib, synth
from posepaf.api import PosePostProcessor

B = 64
L = _lib.load()
post = PosePostProcessor(max_batch=B, max_h=128, max_w=128, max_peaks_per_part=64)
post.set_mode(1)          # K_A, K_B, k_assemble_wave as three launches: the assembly's stamps are written last
for P in (2, 6, 15, 30, 45):
    nets = np.stack([synth.make_net_output(P, 500 + i, dtype=np.float16) for i in range(16)])
    dev = torch.from_numpy(np.concatenate([nets] * (B // 16))).cuda()
    st = torch.zeros(30 * B * 8, dtype=torch.int64, device="cuda")      # K_B stamps 30*B workgroups into the same buffer
    L.pp_debug_set_stamps(C.c_void_p(st.data_ptr()))
    rec = post.process(dev, 512)
    torch.cuda.synchronize()
    L.pp_debug_set_stamps(None)
    s = st.cpu().numpy().reshape(-1, 8)[:B]
    d = np.diff(s[:, :4], axis=1) / 1e3
    print(f"P={P:2d} conns/img {rec['n_connections'].mean():6.1f} | init {d[:, 0].mean():6.1f}  limbs {d[:, 1].mean():6.1f} (max {d[:, 1].max():6.1f})  "
          f"records {d[:, 2].mean():5.1f} kcyc | classification {s[:, 7].mean() / 1e3:6.1f} kcyc; one-by-one conns/img {s[:, 4].mean():5.1f} "
          f"in {s[:, 5].mean() / 1e3:6.1f} kcyc; id-sum merges {int(s[:, 6].sum())}", flush=True)
