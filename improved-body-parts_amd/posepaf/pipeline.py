"""End-to-end inference pipeline on one GPU: uint8 images resident in HBM -> person records.

    pre-process (A0)  ->  IMHN forward (A1: fused convolution kernels of libposepaf.so, posepaf/fused_model.py)
                      ->  K_A (peaks) / K_B (limb matching + the person assembly streaming under it) (A2..A7, HIP)

The batched engine both bench.py and evaluate.py run on is posepaf/engine.py; this module keeps the simple, eager,
one-call-per-batch form (tests, demo_image.py, the original path).

Replaces the per-image serial loop of evaluate.py:262-267 + process() :70-129 by batched device work: the
network output never leaves HBM (the reference copies it to the host at utils/parse_skeletons.py:80).
"""
from __future__ import annotations

import torch

from . import skeleton as sk
from .api import PosePostProcessor, records_to_numpy
from .fused_model import to_planes


def preprocess_batch(images_u8: torch.Tensor, flip: bool = True, dtype=torch.float32) -> torch.Tensor:
    """utils/parse_skeletons.py:52-73 + utils/util.py:44-65 for a batch of equally sized images (scale 1):
    uint8 BGR (B,H,W,3) -> pad bottom/right to a multiple of 64 with 128 -> /255 -> NHWC in [0,1], each image followed
    by the W-mirror of the padded image -> (2B,Hp,Wp,3).  One HIP kernel (pp_preprocess_u8) on the GPU.
    `np.float32(img / 255)` is a float64 division rounded to float32; float32(x)/255 in float32 is the same
    correctly rounded quotient for every x in 0..255 (checked exhaustively in tests/test_pipeline_cpu.py)."""
    B, H, W, _ = images_u8.shape
    Hp = -(-H // sk.MAX_DOWNSAMPLE) * sk.MAX_DOWNSAMPLE
    Wp = -(-W // sk.MAX_DOWNSAMPLE) * sk.MAX_DOWNSAMPLE
    if images_u8.is_cuda and dtype in (torch.float16, torch.float32):
        import ctypes as C
        from . import _lib
        images_u8 = images_u8.contiguous()
        out = torch.empty((B * (2 if flip else 1), Hp, Wp, 3), dtype=dtype, device=images_u8.device)
        rc = _lib.load().pp_preprocess_u8(C.c_void_p(images_u8.data_ptr()), C.c_void_p(out.data_ptr()),
                                          _lib.PP_F16 if dtype == torch.float16 else _lib.PP_F32, B, H, W, sk.MAX_DOWNSAMPLE,
                                          sk.PAD_VALUE, int(flip),
                                          C.c_void_p(torch.cuda.current_stream(images_u8.device).cuda_stream))
        _lib.check(rc)
        return out
    # host tensors (unit tests of the formula only)
    x = torch.full((B, Hp, Wp, 3), float(sk.PAD_VALUE), dtype=torch.float32)
    x[:, :H, :W] = images_u8.to(torch.float32)
    x = (x / 255.0).to(dtype)
    if not flip:
        return x
    return torch.stack((x, x.flip(2)), dim=1).reshape(-1, Hp, Wp, 3)


class PosePipeline:
    def __init__(self, model: torch.nn.Module, post: PosePostProcessor, dtype=torch.float16, flip: bool = True):
        self.model = model
        self.post = post
        self.dtype = dtype
        self.flip = flip

    @torch.no_grad()
    def forward_maps(self, images_u8: torch.Tensor) -> torch.Tensor:
        """-> (B, 2|1, 50, H/4, W/4) last-stage scale-0 output (utils/parse_skeletons.py:76-80 `[-1][0]`)."""
        x = preprocess_batch(images_u8, self.flip, self.dtype)
        out = self.model(x)
        maps = out[-1][0] if isinstance(out, (list, tuple)) else out
        ns = 2 if self.flip else 1
        return to_planes(maps).view(-1, ns, sk.NUM_CH, maps.shape[-2], maps.shape[-1])

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
