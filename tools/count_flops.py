#!/usr/bin/env python3
"""FLOPs (2 x MAC, convolutions + linear layers) that one forward of the INFERENCE model actually executes, counted on
the meta device (no arithmetic): the fused model skips the last stage's coarse-scale feature / prediction heads, which the
reference module computes and discards (utils/parse_skeletons.py:80 reads only [-1][0]).  bench.py's MFMA roofline uses
the executed number, not the reference module's 529.4 GFLOP."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "improved-body-parts_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def count(fused: bool, size: int = 512) -> float:
    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from posepaf.fused_model import FusedIMHN
    total = [0.0]
    real_conv, real_lin = F.conv2d, F.linear

    def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        y = real_conv(x, w, b, stride, padding, dilation, groups)
        total[0] += 2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        return y

    def linear(x, w, b=None):
        y = real_lin(x, w, b)
        total[0] += 2.0 * y.numel() * w.shape[1]
        return y

    F.conv2d, F.linear = conv2d, linear
    torch.nn.functional.conv2d, torch.nn.functional.linear = conv2d, linear
    try:
        net = NetworkEval(TrainingOpt(), GetConfig("Canonical"), bn=True).eval()
        m = (FusedIMHN.from_network(net).eval() if fused else net).to("meta")
        with torch.no_grad():
            m(torch.empty((1, size, size, 3), device="meta"))
    finally:
        F.conv2d, F.linear = real_conv, real_lin
    return total[0]


if __name__ == "__main__":
    ref, fus = count(False), count(True)
    print(f"reference module : {ref / 1e9:.2f} GFLOP per 512x512 forward")
    print(f"inference model  : {fus / 1e9:.2f} GFLOP per 512x512 forward ({100 * (1 - fus / ref):.2f} % skipped)")
