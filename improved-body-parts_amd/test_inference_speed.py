#!/usr/bin/env python3
"""Network-only forward speed (reference test_inference_speed.py:91-120): the 4-stage IMHN on batches of 512 x 512 images,
one `torch.cuda.synchronize()` per batch, the reference's log line per batch and its running average --

    ==================>Test: [651/7497]	Time 0.435 (0.445)	Speed 36.740 (35.933)

(README.md:85 quotes 38.5 images/s on a 2080 Ti from this loop).  The reference feeds the training DataLoader's images; offline
there is no dataset, so the batches are random normalised images resident on the device (what the network computes does not
depend on their content).  The model is the BN-folded fp16 channels-last `FusedIMHN` on this library's convolution kernels
(`--plain`: the `nn.Module` on PyTorch-ROCm), replayed from a HIP graph unless `--no_graph`.  The reference's apex flags are
accepted and ignored (the precision is fp16 throughout, as under its opt-level O1 / O2).

    python test_inference_speed.py [--batch 8] [--iters 50] [-p checkpoint.pth] [--plain] [--no_graph] [--json]
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


class AverageMeter:
    """test_inference_speed.py:52-65"""

    def __init__(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--checkpoint_path", "-p", default=None, help="reference checkpoint (.pth with a 'weights' entry)")
    ap.add_argument("--batch", type=int, default=8, help="images per batch (config/config.py batch_size of the reference: 8)")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3, help="untimed batches (kernel-choice tuning happens in the first)")
    ap.add_argument("--size", type=int, nargs=2, default=[512, 512], metavar=("H", "W"))
    ap.add_argument("--plain", action="store_true", help="the nn.Module on PyTorch-ROCm instead of the fused model")
    ap.add_argument("--no_graph", action="store_true")
    ap.add_argument("--json", action="store_true", help="one JSON summary line at the end")
    for flag in ("--opt-level", "--keep-batchnorm-fp32", "--loss-scale", "--output", "--max_grad_norm"):   # reference flags, unused
        ap.add_argument(flag, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--resume", "-r", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("test_inference_speed.py needs the GPU (there is no CPU path)")
    from posepaf import fused_model as fm
    dev = torch.device("cuda", 0)
    if a.checkpoint_path:
        from config.config import GetConfig, TrainingOpt
        from models.posenet import NetworkEval
        net = NetworkEval(TrainingOpt(), GetConfig(TrainingOpt.config_name), bn=True).eval()
        net.load_state_dict(torch.load(a.checkpoint_path, map_location="cpu", weights_only=True)["weights"])   # :70-71
        model = net if a.plain else fm.FusedIMHN.from_network(net).eval()
        model = model.to(device=dev, dtype=torch.float16).to(memory_format=torch.channels_last)
    else:
        model = fm.build_inference_model(dev, fused=not a.plain)
    h, w = a.size
    g = torch.Generator(device="cpu").manual_seed(0)
    images = torch.rand(a.batch, h, w, 3, generator=g).to(dev).half()          # the loader's normalised NHWC images (:95-97)
    with torch.no_grad():
        if not a.plain and fm.load_table():
            print("kernel-choice table", fm.table_hash(), "loaded", file=sys.stderr)
        for _ in range(max(1, a.warmup)):
            model(images)
        torch.cuda.synchronize()
        run = model
        if not a.no_graph:
            run = fm.GraphedForward(model, images, warmup=1)
        batch_time = AverageMeter()
        torch.cuda.synchronize()
        end = time.time()
        for i in range(a.iters):
            run(images)
            torch.cuda.synchronize()            # :104: the reference times each batch to its completion
            batch_time.update(time.time() - end)
            end = time.time()
            print("==================>Test: [{0}/{1}]\tTime {bt.val:.3f} ({bt.avg:.3f})\tSpeed {2:.3f} ({3:.3f})\t".format(
                i, a.iters, a.batch / batch_time.val, a.batch / batch_time.avg, bt=batch_time))
    if a.json:
        print(json.dumps({"metric": "network-only forward images/sec", "value": a.batch / batch_time.avg, "batch": a.batch,
                          "size": [h, w], "iters": a.iters, "model": "nn.Module (PyTorch-ROCm)" if a.plain else "FusedIMHN",
                          "launch": "eager" if a.no_graph else "hipGraph replay", "dtype": "f16",
                          "conv_table": None if a.plain else fm.table_hash()}))
    return a.batch / batch_time.avg


if __name__ == "__main__":
    main()
