"""End-to-end inference pipeline on one GPU: uint8 images resident in HBM -> person records.

    pre-process (A0)  ->  IMHN forward (A1, PyTorch-ROCm convolutions)  ->  K_A / K_B / K_C (A2..A7, HIP)

Replaces the per-image serial loop of evaluate.py:262-267 + process() :70-129 by batched device work: the
network output never leaves HBM (the reference copies it to the host at utils/parse_skeletons.py:80).
"""
from __future__ import annotations

import torch

from . import skeleton as sk
from .api import PosePostProcessor, records_to_numpy


def preprocess_batch(images_u8: torch.Tensor, flip: bool = True) -> torch.Tensor:
    """utils/parse_skeletons.py:52-73 for a batch of equally sized images already padded to a multiple of 64:
    uint8 BGR (B,H,W,3) -> float32 NHWC in [0,1], each image followed by its W-mirrored copy -> (2B,H,W,3).
    `np.float32(img / 255)` is a float64 division rounded to float32; float32(x)/255 in float32 is the same
    correctly rounded quotient for every x in 0..255 (checked exhaustively in tests/test_pipeline_cpu.py)."""
    x = images_u8.to(torch.float32) / 255.0
    if not flip:
        return x
    return torch.stack((x, x.flip(2)), dim=1).reshape(-1, *x.shape[1:])


class PosePipeline:
    def __init__(self, model: torch.nn.Module, post: PosePostProcessor, dtype=torch.float16, flip: bool = True):
        self.model = model
        self.post = post
        self.dtype = dtype
        self.flip = flip

    @torch.no_grad()
    def forward_maps(self, images_u8: torch.Tensor) -> torch.Tensor:
        """-> (B, 2|1, 50, H/4, W/4) last-stage scale-0 output (utils/parse_skeletons.py:76-80 `[-1][0]`)."""
        x = preprocess_batch(images_u8, self.flip).to(self.dtype)
        out = self.model(x)
        maps = out[-1][0] if isinstance(out, (list, tuple)) else out
        ns = 2 if self.flip else 1
        return maps.contiguous().view(-1, ns, sk.NUM_CH, maps.shape[-2], maps.shape[-1])

    @torch.no_grad()
    def run_async(self, images_u8: torch.Tensor, min_img_size: int | None = None, inject: torch.Tensor | None = None):
        """Enqueue one batch; returns the device record buffer (no host sync).
        inject: optional (B,2,50,h,w) maps ADDED to the network output before post-processing.  Used by the
        benchmark only: a randomly initialised network emits no peaks, so realistic post-processing load is
        supplied as synthetic ground-truth-style maps while the forward itself still runs in full."""
        maps = self.forward_maps(images_u8)
        if inject is not None:
            maps = maps + inject
        mis = images_u8.shape[1] if min_img_size is None else min_img_size
        return self.post.process_async(maps, mis, self.flip)

    def __call__(self, images_u8: torch.Tensor, min_img_size: int | None = None):
        return records_to_numpy(self.run_async(images_u8, min_img_size))


def smoke_forward(post: PosePostProcessor) -> None:
    """One tiny forward (64x64 image) of the real architecture feeding the HIP kernels."""
    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from .model_init import deterministic_init
    model = NetworkEval(TrainingOpt(), GetConfig("Canonical"), bn=True).eval()
    deterministic_init(model, 7)
    model = model.to("cuda:0").half()
    pipe = PosePipeline(model, post)
    img = torch.randint(0, 256, (1, 64, 64, 3), dtype=torch.uint8, device="cuda:0")
    rec = pipe(img)
    assert rec.shape[0] == 1 and int(rec[0]["n_humans"]) >= 0
