import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "improved-body-parts_amd")):
    sys.path.insert(0, p)
import torch
from posepaf import fused_model as fm
f = fm.FConv(torch.nn.Conv2d(256, 256, 1, bias=True), None, False).cuda().half()
n = 128
x = torch.randn(n, 256, 128, 128, device="cuda").half().contiguous(memory_format=torch.channels_last)
res = torch.randn_like(x).contiguous(memory_format=torch.channels_last)
other = torch.randn_like(x).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    y, y2 = f.forward_dual(x, res, other)
for k, v in fm._conv_timing.items():
    print(k, {a: round(b, 3) for a, b in v.items()})
print(fm._conv_choice)
