"""The reference's original (non-refactored) inference path with a real scale search (BASELINE config 5):

    predict (utils/parse_skeletons.py:180-283)  ->  find_peaks (:286-321)  ->  find_connections / find_humans (:324-600)

for a batch of equally sized images, everything on the GPU: per scale the uint8 images are resized (bicubic), padded,
normalised and mirrored, run through the network, flip-averaged, up-sampled x4, cropped, resized to the image size and
accumulated in float64 maps that stay in HBM (105 MB per 512x512 image); peaks, matching and assembly then run at image
resolution.  `multiplier` is the list predict builds at :186 (the reference then hard-codes [1.] at :188; config 5 asks
for three scales)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import skeleton as sk
from ._lib import RECORD_BYTES
from .api import PosePostProcessor, records_to_numpy
from .pipeline import preprocess_batch


def _p(t):
    return C.c_void_p(t.data_ptr())


def scaled_size(h: int, w: int, scale: float):
    """cv2.resize(image, (0, 0), fx=scale, fy=scale): dsize = round(size * scale)"""
    return int(round(h * scale)), int(round(w * scale))


def resize_images_u8(images_u8: torch.Tensor, scale: float) -> torch.Tensor:
    B, H, W, _ = images_u8.shape
    if scale == 1.0:
        return images_u8
    dh, dw = scaled_size(H, W, scale)
    out = torch.empty((B, dh, dw, 3), dtype=torch.uint8, device=images_u8.device)
    st = C.c_void_p(torch.cuda.current_stream(images_u8.device).cuda_stream)
    _lib.check(_lib.load().pp_resize_u8_cubic(_p(images_u8.contiguous()), _p(out), B, H, W, dh, dw, 1.0 / scale, 1.0 / scale, st))
    return out


class OriginalPathProcessor:
    def __init__(self, post: PosePostProcessor, img_h: int, img_w: int, max_batch: int, device=None):
        if post.maxp > 64:
            raise _lib.PosePafError("the original path needs max_peaks_per_part <= 64 (float64 tables in LDS)")
        self.post, self.H, self.W, self.B = post, img_h, img_w, max_batch
        dev = device or torch.device("cuda", post.device)
        self.heat_acc = torch.zeros((max_batch, sk.NUM_HEAT, img_h, img_w), dtype=torch.float64, device=dev)
        self.paf_acc = torch.zeros((max_batch, sk.NUM_LIMB, img_h, img_w), dtype=torch.float64, device=dev)
        self.mask = torch.empty((max_batch, sk.NUM_PART, img_h, img_w), dtype=torch.uint8, device=dev)
        self.peaks64 = torch.empty((max_batch, sk.NUM_PART, post.maxp, 4), dtype=torch.float64, device=dev)
        self.records = torch.empty(max_batch * RECORD_BYTES, dtype=torch.uint8, device=dev)
        self._scratch = {}

    def reset(self):
        self.heat_acc.zero_()
        self.paf_acc.zero_()

    def accumulate(self, maps: torch.Tensor, pad_down: int, pad_right: int, n_scales: int, flip: bool = True):
        """maps: (B, 2|1, 50, h, w) network output of ONE scale (padded input); adds it to the accumulators."""
        B, _, _, h, w = maps.shape
        key = (B, h, w)
        if key not in self._scratch:
            self._scratch[key] = (torch.empty((B, sk.NUM_CH, h, w), dtype=torch.float32, device=maps.device),
                                  torch.empty((B, sk.NUM_CH, 4 * h, 4 * w), dtype=torch.float32, device=maps.device))
        planar, up = self._scratch[key]
        code = _lib.PP_F16 if maps.dtype == torch.float16 else _lib.PP_F32
        st = C.c_void_p(torch.cuda.current_stream(maps.device).cuda_stream)
        _lib.check(_lib.load().pp_original_accumulate(self.post.ctx, B, _p(maps), code, h, w, int(flip), pad_down, pad_right,
                                                      self.H, self.W, n_scales, _p(planar), _p(up), _p(self.heat_acc),
                                                      _p(self.paf_acc), st), self.post.ctx)

    def finish(self, batch: int, thre1: float = 0.1) -> torch.Tensor:
        st = C.c_void_p(torch.cuda.current_stream(self.heat_acc.device).cuda_stream)
        _lib.check(_lib.load().pp_original_finish(self.post.ctx, batch, self.H, self.W, float(thre1), _p(self.heat_acc),
                                                  _p(self.paf_acc), _p(self.mask), _p(self.peaks64), _p(self.records), st),
                   self.post.ctx)
        return self.records[: batch * RECORD_BYTES]

    @torch.no_grad()
    def run(self, model, images_u8: torch.Tensor, multiplier, dtype=torch.float16, thre1: float = 0.1) -> np.ndarray:
        """images (B, H, W, 3) uint8 on the GPU -> records (float coordinates, PP_ST_FLOAT_COORDS)."""
        B = images_u8.shape[0]
        self.reset()
        for scale in multiplier:
            scaled = resize_images_u8(images_u8, float(scale))
            sh, sw = scaled.shape[1:3]
            x = preprocess_batch(scaled, True, dtype)                       # pad to /64, /255, mirror
            ph, pw = x.shape[1:3]
            out = model(x)
            maps = (out[-1][0] if isinstance(out, (list, tuple)) else out).contiguous()
            maps = maps.view(B, 2, sk.NUM_CH, maps.shape[-2], maps.shape[-1])
            self.accumulate(maps, ph - sh, pw - sw, len(multiplier))
        return records_to_numpy(self.finish(B, thre1))


def record_float_coords(rec):
    """x / y of a PP_ST_FLOAT_COORDS record as float32 arrays (n_humans, 18)."""
    n = int(rec["n_humans"])
    return rec["humans"]["x"][:n].view(np.float32), rec["humans"]["y"][:n].view(np.float32)
