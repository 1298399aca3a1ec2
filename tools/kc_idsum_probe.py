#!/usr/bin/env python3
"""Confirms that the hand-built id-summing merge scene takes K_C's table-abandon branch (diagnostic stamps); GPU only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
from posepaf import _lib, synth
from posepaf.api import PosePostProcessor

L = _lib.load()
post = PosePostProcessor(max_batch=1, max_h=128, max_w=128, max_peaks_per_part=64)
dev = torch.from_numpy(synth.make_id_sum_merge_scene()).cuda()[None]
st = torch.zeros(30 * 8, dtype=torch.int64, device="cuda")
L.pp_debug_set_stamps(C.c_void_p(st.data_ptr()))
rec = post.process(dev, 512, flip=False)[0]
torch.cuda.synchronize()
L.pp_debug_set_stamps(None)
s = st.cpu().numpy().reshape(-1, 8)
print("humans", int(rec["n_humans"]), "nose id", int(rec["humans"]["peak_id"][0, 0]), "| one-by-one connections", int(s[0, 4]),
      "id-summing merges", int(s[0, 6]))
assert s[0, 6] == 1
