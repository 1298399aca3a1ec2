"""Inference glue with the reference's function names (utils/parse_skeletons.py), backed by the HIP path.

    predict_refactor(image, model, test_cfg, model_cfg, path, flip_avg=True, config=None) -> (heat, paf)   :28-103
    find_peaks_refactor(param, img) -> (n, 2) [x, y]                                                        :106-119
    heatmap_nms(heatmaps, upsample_factor=4, bool_refine_center=True) -> list[18] of (n, 4)                 :126-176

All three run on the GPU (pre-processing kernel -> model -> pp_flip_average / pp_nms_batch); host arrays are only the
inputs/outputs the reference's signatures prescribe.  For throughput use posepaf.pipeline.PosePipeline, which keeps the
network output in HBM and never builds these intermediate arrays.

`find_connections` / `find_humans` (the pure-Python twins, :324-600) follow DIFFERENT rules from the C++ `pafprocess`
path this repository reproduces; they are not provided yet and raise NotImplementedError pointing
at utils.pafprocess.
"""
import ctypes as C

import numpy as np
import torch

from posepaf import _lib
from posepaf import skeleton as sk
from posepaf.pipeline import preprocess_batch

NUM_KEYPOINTS = 18
NUM_HEATMAPS = NUM_KEYPOINTS + 2
NUM_PAFS = 30

_post_cache = {}


def _post(h, w):
    from posepaf.api import PosePostProcessor
    key = (h, w)
    if key not in _post_cache:
        _post_cache[key] = PosePostProcessor(max_batch=1, max_h=h, max_w=w, max_peaks_per_part=128)
    return _post_cache[key]


def predict_refactor(image, model, test_cfg, model_cfg, input_image_path, flip_avg=True, config=None):
    """BGR uint8 (H, W, 3) -> (heatmap_avg (h, w, 20), paf_avg (h, w, 30)) float32, h = padded H / 4.
    Scale search is fixed to [1.0] as in the reference (:36); rotation_search must be [0]."""
    if any(a != 0 for a in test_cfg.get("rotation_search", [0.0])):
        raise NotImplementedError("rotation_search != 0 is not supported (the reference's default is 0)")
    dev = next(model.parameters()).device
    dtype = next(model.parameters()).dtype
    img = torch.from_numpy(np.ascontiguousarray(image)).to(dev)[None]
    with torch.no_grad():
        x = preprocess_batch(img, True, dtype if dtype in (torch.float16, torch.float32) else torch.float32)
        out = model(x)
        maps = (out[-1][0] if isinstance(out, (list, tuple)) else out).contiguous()   # (2, 50, h, w)
    h, w = maps.shape[-2:]
    heat = torch.empty((h, w, NUM_HEATMAPS), dtype=torch.float32, device=dev)
    paf = torch.empty((h, w, NUM_PAFS), dtype=torch.float32, device=dev)
    code = _lib.PP_F16 if maps.dtype == torch.float16 else _lib.PP_F32
    if maps.dtype not in (torch.float16, torch.float32):
        maps, code = maps.float(), _lib.PP_F32
    src = maps if flip_avg else maps[:1].contiguous()
    _lib.check(_lib.load().pp_flip_average(C.c_void_p(src.data_ptr()), code, 1, h, w, int(bool(flip_avg)),
                                           C.c_void_p(heat.data_ptr()), C.c_void_p(paf.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return heat.cpu().numpy(), paf.cpu().numpy()


def _planar_from_heat(heatmaps):
    hm = np.ascontiguousarray(heatmaps, dtype=np.float32)
    if hm.ndim != 3 or hm.shape[2] < NUM_KEYPOINTS:
        raise ValueError("heatmaps must be (h, w, >=18)")
    h, w = hm.shape[:2]
    net = torch.zeros((1, 1, sk.NUM_CH, h, w), dtype=torch.float32, device="cuda")
    net[0, 0, sk.NUM_LIMB:sk.NUM_LIMB + NUM_KEYPOINTS] = torch.from_numpy(hm[:, :, :NUM_KEYPOINTS]).cuda().permute(2, 0, 1)
    return net, h, w


def heatmap_nms(heatmaps, upsample_factor=1., bool_refine_center=True):
    """list over the 18 parts of float32 (n, 4) rows [x, y, score, peak_id] -- utils/parse_skeletons.py:126-176."""
    if int(upsample_factor) != sk.STRIDE:
        raise NotImplementedError("only upsample_factor == stride == 4 is implemented (what evaluate.py:76 passes)")
    net, h, w = _planar_from_heat(heatmaps)
    jl = _post(h, w).nms(net, flip=False, refine=bool(bool_refine_center))[0]
    return [jl[jl[:, 4] == part][:, :4].copy() for part in range(NUM_KEYPOINTS)]


def find_peaks_refactor(param, img):
    """(n, 2) integer [x, y] of the plus-shaped local maxima above `param` -- :106-119 (param must be 0.1, the only
    value the reference ever passes, :139)."""
    if abs(float(param) - sk.NMS_THRESHOLD) > 1e-12:
        raise NotImplementedError("threshold is fixed at 0.1 like heatmap_nms's hard-coded call (:139)")
    m = np.ascontiguousarray(img, dtype=np.float32)
    hm = np.zeros(m.shape + (NUM_KEYPOINTS,), np.float32)
    hm[:, :, 0] = m
    net, h, w = _planar_from_heat(hm)
    jl = _post(h, w).nms(net, flip=False, refine=False)[0]
    xy = (jl[jl[:, 4] == 0][:, :2] + 0.5) / sk.STRIDE - 0.5       # undo compute_resized_coords (:122-123), exact
    return np.rint(xy).astype(np.intp)


def find_connections(*args, **kwargs):
    raise NotImplementedError("the pure-Python matching (parse_skeletons.py:324-410) is not reproduced; use "
                              "utils.pafprocess.pafprocess.process_paf (C++ semantics) or posepaf.api.PosePostProcessor")


def find_humans(*args, **kwargs):
    raise NotImplementedError("the pure-Python assembly (parse_skeletons.py:413-600) is not reproduced; use "
                              "utils.pafprocess.pafprocess.process_paf (C++ semantics) or posepaf.api.PosePostProcessor")
