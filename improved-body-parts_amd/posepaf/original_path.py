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
from .fused_model import to_planes


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
    """Accumulators at image resolution + the kernels behind them.  accumulate() only REGISTERS a scale; the arithmetic of all
    registered scales runs in ONE launch (pp_original_accumulate_all: the accumulators are written once, never re-read) when the
    accumulators are first needed -- by finish(), or by reading .heat_acc / .paf_acc."""

    def __init__(self, post: PosePostProcessor, img_h: int, img_w: int, max_batch: int, device=None):
        if post.maxp > 64:
            raise _lib.PosePafError("the original path needs max_peaks_per_part <= 64 (float64 tables in LDS)")
        self.post, self.H, self.W, self.B = post, img_h, img_w, max_batch
        dev = device or torch.device("cuda", post.device)
        self._heat = torch.zeros((max_batch, sk.NUM_HEAT, img_h, img_w), dtype=torch.float64, device=dev)
        self._paf = torch.zeros((max_batch, sk.NUM_LIMB, img_h, img_w), dtype=torch.float64, device=dev)
        self.mask = torch.empty((max_batch, sk.NUM_PART, img_h, img_w), dtype=torch.uint8, device=dev)
        self.peaks64 = torch.empty((max_batch, sk.NUM_PART, post.maxp, 4), dtype=torch.float64, device=dev)
        self.records = torch.empty(max_batch * RECORD_BYTES, dtype=torch.uint8, device=dev)
        self._scratch = {}
        self._pending = []        # [(maps, pad_down, pad_right, n_scales, flip)] registered, not yet accumulated
        self._need_zero = False   # reset() was called and nothing has written the accumulators since
        self.fused = True         # False: always the per-scale chain (pp_original_accumulate), for A/B measurements

    @property
    def heat_acc(self) -> torch.Tensor:
        self._flush()
        return self._heat

    @property
    def paf_acc(self) -> torch.Tensor:
        self._flush()
        return self._paf

    def reset(self):
        self._pending = []
        self._need_zero = True

    def accumulate(self, maps: torch.Tensor, pad_down: int, pad_right: int, n_scales: int, flip: bool = True):
        """maps: (B, 2|1, 50, h, w) network output of ONE scale (padded input).  The tensor must stay alive until finish()."""
        self._pending.append((maps, int(pad_down), int(pad_right), int(n_scales), bool(flip)))

    def _chain(self, maps, pad_down, pad_right, n_scales, flip):
        """one scale through the round-2 chain: flip-average -> x4 map -> crop -> resize -> read-modify-write accumulate"""
        B, _, _, h, w = maps.shape
        key = (B, h, w)
        if key not in self._scratch:
            self._scratch[key] = (torch.empty((B, sk.NUM_CH, h, w), dtype=torch.float32, device=maps.device),
                                  torch.empty((B, sk.NUM_CH, 4 * h, 4 * w), dtype=torch.float32, device=maps.device))
        planar, up = self._scratch[key]
        code = _lib.PP_F16 if maps.dtype == torch.float16 else _lib.PP_F32
        st = C.c_void_p(torch.cuda.current_stream(maps.device).cuda_stream)
        _lib.check(_lib.load().pp_original_accumulate(self.post.ctx, B, _p(maps), code, h, w, int(flip), pad_down, pad_right,
                                                      self.H, self.W, n_scales, _p(planar), _p(up), _p(self._heat),
                                                      _p(self._paf), st), self.post.ctx)

    def _flush(self):
        pend, self._pending = self._pending, []
        if not pend:
            if self._need_zero:
                self._heat.zero_()
                self._paf.zero_()
                self._need_zero = False
            return
        maps0, _, _, n_scales, flip = pend[0]
        B = maps0.shape[0]
        same = all(m.shape[0] == B and m.dtype == maps0.dtype and n == n_scales and f == flip and m.is_contiguous()
                   for m, _, _, n, f in pend)
        if self.fused and self._need_zero and same and len(pend) <= 6 and B <= self.B:
            L = _lib.load()
            n = len(pend)
            ptrs = (C.c_void_p * n)(*[m.data_ptr() for m, *_ in pend])
            hs = (C.c_int * n)(*[m.shape[3] for m, *_ in pend])
            ws = (C.c_int * n)(*[m.shape[4] for m, *_ in pend])
            pd = (C.c_int * n)(*[p[1] for p in pend])
            pr = (C.c_int * n)(*[p[2] for p in pend])
            code = _lib.PP_F16 if maps0.dtype == torch.float16 else _lib.PP_F32
            st = C.c_void_p(torch.cuda.current_stream(maps0.device).cuda_stream)
            # n_div of the kernel is the number of scales it is given: only the complete set goes through it
            if n == n_scales:
                rc = L.pp_original_accumulate_all(self.post.ctx, B, n, ptrs, code, hs, ws, int(flip), pd, pr, self.H, self.W,
                                                  _p(self._heat), _p(self._paf), st)
                if rc == 0:
                    self._need_zero = False
                    return
                if rc != -6:   # PP_ERR_UNSUPPORTED (a scale too large for the tiles): fall through to the chain
                    _lib.check(rc, self.post.ctx)
        if self._need_zero:
            self._heat.zero_()
            self._paf.zero_()
            self._need_zero = False
        for m, pdn, prt, n, f in pend:
            self._chain(m, pdn, prt, n, f)

    def finish(self, batch: int, thre1: float = 0.1) -> torch.Tensor:
        self._flush()
        st = C.c_void_p(torch.cuda.current_stream(self._heat.device).cuda_stream)
        _lib.check(_lib.load().pp_original_finish(self.post.ctx, batch, self.H, self.W, float(thre1), _p(self._heat),
                                                  _p(self._paf), _p(self.mask), _p(self.peaks64), _p(self.records), st),
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
            maps = to_planes(out[-1][0] if isinstance(out, (list, tuple)) else out)
            maps = maps.view(B, 2, sk.NUM_CH, maps.shape[-2], maps.shape[-1])
            self.accumulate(maps, ph - sh, pw - sw, len(multiplier))
        return records_to_numpy(self.finish(B, thre1))


def record_float_coords(rec):
    """x / y of a PP_ST_FLOAT_COORDS record as float32 arrays (n_humans, 18)."""
    n = int(rec["n_humans"])
    return rec["humans"]["x"][:n].view(np.float32), rec["humans"]["y"][:n].view(np.float32)
