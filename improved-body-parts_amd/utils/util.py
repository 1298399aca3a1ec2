"""Helpers with the reference's names (utils/util.py): padRightDownCorner (:44-65), keypoint_heatmap_nms (:177-185),
refine_centroid (:188-213).  Colour-map / Gaussian-smoothing utilities of that file are visualisation and out of scope."""
import numpy as np
import torch
import torch.nn.functional as F


def padRightDownCorner(img, stride, padValue):
    """Pad bottom/right so that H and W become multiples of `stride`; returns (padded, [up, left, down, right])."""
    h, w = img.shape[0], img.shape[1]
    pad = [0, 0, 0 if h % stride == 0 else stride - (h % stride), 0 if w % stride == 0 else stride - (w % stride)]
    out = np.full((h + pad[2], w + pad[3]) + img.shape[2:], padValue, dtype=img.dtype)
    out[:h, :w] = img
    return out, pad


def keypoint_heatmap_nms(heat, kernel=3, thre=0.1):
    """3x3 max-pool NMS on a (1, C, H, W) tensor (stays on the tensor's device): keep heat where it equals the
    reflect-padded window maximum and is >= thre.  A float32 / float16 device tensor with up to 18 channels goes through the
    HIP peak kernel's 3x3 mode (pp_nms_batch_ex, nms_mode 1: the same comparison rules) whenever the map fits its LDS
    tile and no channel has more peaks than the kernel keeps; everything else through the torch expression below."""
    if kernel == 3 and heat.is_cuda and heat.dim() == 4 and heat.shape[0] == 1 and heat.shape[1] <= 18 and \
            heat.dtype in (torch.float16, torch.float32):
        out = _hip_keypoint_nms(heat, float(thre))
        if out is not None:
            return out
        _warn_once("keypoint_heatmap_nms: the HIP peak kernel did not take this map (too large for its LDS tile, or a channel "
                   "with more than 128 peaks) -- the torch max-pool expression runs instead")
    pad = (kernel - 1) // 2
    hmax = F.max_pool2d(F.pad(heat, (pad, pad, pad, pad), mode="reflect"), (kernel, kernel), stride=1, padding=0)
    keep = (hmax == heat).float() * (heat >= thre).float()
    return heat * keep


def refine_centroid(scorefmp, anchor, radius):
    """(2r+1)^2 weighted centroid around an integer peak; border peaks are returned unrefined with the raw score,
    otherwise the score is the box MEAN (utils/util.py:200-213, including its row/column grid convention)."""
    x_c, y_c = anchor
    x_min, x_max, y_min, y_max = x_c - radius, x_c + radius + 1, y_c - radius, y_c + radius + 1
    if y_max > scorefmp.shape[0] or y_min < 0 or x_max > scorefmp.shape[1] or x_min < 0:
        return anchor + (scorefmp[y_c, x_c],)
    box = scorefmp[y_min:y_max, x_min:x_max]
    x_grid, y_grid = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    offset_x = (box * x_grid).sum() / box.sum()
    offset_y = (box * y_grid).sum() / box.sum()
    return (x_c + offset_x, y_c + offset_y) + (box.mean(),)


_nms_ctx = {}
_warned = set()


def _warn_once(msg):
    """a device tensor that leaves the HIP path says so once (POSEPAF_STRICT=1: raises instead)"""
    import os
    import warnings
    if os.environ.get("POSEPAF_STRICT", "0") == "1":
        raise RuntimeError(msg)
    if msg not in _warned:
        _warned.add(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _hip_keypoint_nms(heat, thre):
    """heat * keep from the peak list of K_A's 3x3 / >= mode; None when the HIP path does not take the map (too large for the
    LDS tile, or a channel with more than 128 peaks)."""
    from posepaf import _lib
    from posepaf.api import PosePostProcessor
    _, c, h, w = heat.shape
    try:
        if (h, w) not in _nms_ctx:
            _nms_ctx[(h, w)] = PosePostProcessor(max_batch=1, max_h=h, max_w=w, max_peaks_per_part=128, device=heat.device.index or 0)
        post = _nms_ctx[(h, w)]
        net = torch.zeros((1, 1, 50, h, w), dtype=heat.dtype, device=heat.device)
        net[0, 0, 30:30 + c] = heat[0]
        rows = post.nms_ex(net, flip=False, nms_mode=1, threshold=thre, refine_mode=3)[0]   # integer coordinates, raw score
    except _lib.PosePafError:
        return None
    rows = rows[rows[:, 4] < c]
    out = torch.zeros_like(heat)
    if len(rows):
        idx = torch.from_numpy(rows[:, [4, 1, 0]].astype("int64")).to(heat.device)
        out[0, idx[:, 0], idx[:, 1], idx[:, 2]] = heat[0, idx[:, 0], idx[:, 1], idx[:, 2]]
    return out
