"""Inference glue with the reference's function names (utils/parse_skeletons.py), backed by the HIP path.

    predict_refactor(image, model, test_cfg, model_cfg, path, flip_avg=True, config=None) -> (heat, paf)   :28-103
    find_peaks_refactor(param, img) -> (n, 2) [x, y]                                                        :106-119
    heatmap_nms(heatmaps, upsample_factor=4, bool_refine_center=True) -> list[18] of (n, 4)                 :126-176

All three run on the GPU (pre-processing kernel -> model -> pp_flip_average / pp_nms_batch); host arrays are only the
inputs/outputs the reference's signatures prescribe.  For throughput use posepaf.pipeline.PosePipeline, which keeps the
network output in HBM and never builds these intermediate arrays.

    predict(image, model, test_cfg, model_cfg, path, flip_avg=True, config=None) -> (heatmap_avg, paf_avg)  :180-283
    find_peaks(heatmap_avg, test_cfg) -> list[18] of [(x, y, score, id), ...]                               :286-321

the original (non-refactored) path of evaluate.py:81-84 at IMAGE resolution, thin callers of
posepaf.original_path.OriginalPathProcessor (bicubic x4, crop, resize to the image, float64 accumulation; 3x3 / >= thre1
NMS + refine_centroid) -- all on the GPU.

    find_connections(all_peaks, paf_avg, img_height, test_cfg, joint2limb_pairs) -> (connected_limbs, special_limb)  :324-410
    find_humans(connected_limbs, special_limb, joint_list, test_cfg, joint2limb_pairs) -> (persons, candidates)     :413-600

The last two are the pure-Python twins, whose rules differ from the C++ `pafprocess` path; both
rule sets are implemented as kernels.
"""
import ctypes as C

import numpy as np
import torch

from posepaf import _lib
from posepaf import skeleton as sk
from posepaf.pipeline import preprocess_batch
from posepaf.fused_model import to_planes

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
        maps = to_planes(out[-1][0] if isinstance(out, (list, tuple)) else out)   # (2, 50, h, w)
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


_orig_cache = {}


def _original(img_h, img_w):
    """one OriginalPathProcessor (accumulators at image resolution) per image size"""
    from posepaf.original_path import OriginalPathProcessor
    key = (img_h, img_w)
    if key not in _orig_cache:
        hp = -(-img_h // sk.MAX_DOWNSAMPLE) * sk.MAX_DOWNSAMPLE // sk.STRIDE
        wp = -(-img_w // sk.MAX_DOWNSAMPLE) * sk.MAX_DOWNSAMPLE // sk.STRIDE
        from posepaf.api import PosePostProcessor
        post = PosePostProcessor(max_batch=1, max_h=hp, max_w=wp, max_peaks_per_part=64)
        _orig_cache[key] = OriginalPathProcessor(post, img_h, img_w, 1)
    return _orig_cache[key]


def predict(image, model, test_cfg, model_cfg, input_image_path, flip_avg=True, config=None):
    """utils/parse_skeletons.py:180-283: BGR uint8 (H, W, 3) -> (heatmap_avg (H, W, 20), paf_avg (H, W, 30)) float64 at IMAGE
    resolution.  The scale list is [1.0], as the reference fixes it at :188 (its `scale_search` line :186 is dead);
    `test_cfg["multiplier"]`, when present, overrides it (BASELINE config 5 uses 0.5 / 1.0 / 1.5).  rotation_search must be [0]."""
    if any(a != 0 for a in test_cfg.get("rotation_search", [0.0])):
        raise NotImplementedError("rotation_search != 0 is not supported (the reference's default is 0)")
    from posepaf.original_path import resize_images_u8
    img_h, img_w = image.shape[:2]
    proc = _original(img_h, img_w)
    dev = next(model.parameters()).device
    dtype = next(model.parameters()).dtype
    dtype = dtype if dtype in (torch.float16, torch.float32) else torch.float32
    img = torch.from_numpy(np.ascontiguousarray(image)).to(dev)[None]
    multiplier = [float(m) for m in test_cfg.get("multiplier", [1.0])]
    with torch.no_grad():
        proc.reset()
        for scale in multiplier:
            scaled = resize_images_u8(img, scale)                          # cv2.resize(image, fx=scale), :204
            sh, sw = scaled.shape[1:3]
            x = preprocess_batch(scaled, True, dtype)                      # pad to /64 with 128, /255, mirror (:206-226)
            out = model(x)
            maps = to_planes(out[-1][0] if isinstance(out, (list, tuple)) else out)
            if maps.dtype not in (torch.float16, torch.float32):
                maps = maps.float()
            maps = maps.view(1, 2, sk.NUM_CH, maps.shape[-2], maps.shape[-1])
            if not flip_avg:
                maps = maps[:, :1].contiguous()                            # :236-237 the un-mirrored sample alone
            proc.accumulate(maps, x.shape[1] - sh, x.shape[2] - sw, len(multiplier), flip=bool(flip_avg))
    heat = proc.heat_acc[0].permute(1, 2, 0).contiguous().cpu().numpy()
    paf = proc.paf_acc[0].permute(1, 2, 0).contiguous().cpu().numpy()
    return heat, paf


def find_peaks(heatmap_avg, test_cfg):
    """utils/parse_skeletons.py:286-321: (H, W, >=18) maps at image resolution -> list over the 18 parts of
    [(x, y, score, id), ...]: 3x3 / >= thre1 NMS (util.keypoint_heatmap_nms) in np.nonzero order, util.refine_centroid with
    radius 2 (x / y fractional, score = box mean), ids running across parts."""
    if int(test_cfg.get("offset_radius", 2)) != 2:
        raise NotImplementedError("offset_radius is fixed at 2 (utils/config:26)")
    hm = np.asarray(heatmap_avg)
    if hm.ndim != 3 or hm.shape[2] < NUM_KEYPOINTS:
        raise ValueError("heatmap_avg must be (H, W, >=18)")
    H, W = hm.shape[:2]
    proc = _original(H, W)
    with torch.no_grad():
        proc.reset()
        # the kernel casts the float64 accumulator to float32 on read, like heatmap_avg.astype(np.float32) at :290
        proc.heat_acc[0, :NUM_KEYPOINTS] = torch.from_numpy(np.ascontiguousarray(hm[:, :, :NUM_KEYPOINTS].transpose(2, 0, 1)))\
            .to(proc.heat_acc.device, torch.float64)
        proc.finish(1, float(test_cfg.get("thre1", 0.1)))
    counts = proc.post.read_part_counts(0)
    if (counts > proc.post.maxp).any():
        raise _lib.PosePafError(f"more than {proc.post.maxp} peaks in one keypoint channel: refused, not truncated")
    rows = proc.peaks64[0].cpu().numpy()                                   # (18, maxp, 4) float64 [x, y, score, -]
    all_peaks, pid = [], 0
    for part in range(NUM_KEYPOINTS):
        n = int(counts[part])
        all_peaks.append([(float(r[0]), float(r[1]), float(r[2]), pid + i) for i, r in enumerate(rows[part, :n])])
        pid += n
    return all_peaks


_ini = sk.default_test_cfg()


def _check_cfg(test_cfg):
    for k in ("thre2", "connect_ration", "mid_num", "len_rate", "connection_tole"):
        if k in test_cfg and float(test_cfg[k]) != float(_ini[k]):
            raise NotImplementedError(f"test_cfg[{k!r}] = {test_cfg[k]}: only the INI defaults (utils/config:17-25) are built in")
    if int(test_cfg.get("remove_recon", 0)) != 0:
        raise NotImplementedError("remove_recon != 0 is not supported (reference default 0)")


def _joint_rows(all_peaks):
    rows = [tuple(float(v) for v in pk[:4]) + (float(part),) for part, pks in enumerate(all_peaks) for pk in pks]
    return np.asarray(rows, np.float32).reshape(-1, 5)


def _ctx():
    """context for the host-form twins: 64 peaks per part (the float64 person table + connections must fit LDS)"""
    from posepaf.api import PosePostProcessor
    if "py" not in _post_cache:
        _post_cache["py"] = PosePostProcessor(max_batch=1, max_h=sk.BOXSIZE // sk.STRIDE, max_w=sk.BOXSIZE // sk.STRIDE,
                                              max_peaks_per_part=64)
    return _post_cache["py"]


def find_connections(all_peaks, paf_avg, img_height, test_cfg, joint2limb_pairs):
    """utils/parse_skeletons.py:324-410 on the GPU (k_limb_connect_py_hwc): -> (connected_limbs, special_limb) with
    connected_limbs[k] an (n, 6) float64 array [src_peak_id, dst_peak_id, score, i, j, limb_len] ([] for special limbs)."""
    _check_cfg(test_cfg)
    if [tuple(p) for p in np.asarray(joint2limb_pairs).tolist()] != [tuple(p) for p in sk.LIMB_PAIRS]:
        raise NotImplementedError("joint2limb_pairs must be the canonical 30 limbs (config/config.py:114-121)")
    post = _ctx()
    jl = _joint_rows(all_peaks)
    paf = np.ascontiguousarray(paf_avg, dtype=np.float32)
    conns = np.zeros((NUM_PAFS, post.maxp, 6), np.float64)
    counts = np.zeros(NUM_PAFS, np.int32)
    special = np.zeros(NUM_PAFS, np.int32)
    L = _lib.load()
    _lib.check(L.pp_py_find_connections_host(post.ctx, jl.ctypes.data_as(C.POINTER(C.c_float)), len(jl),
                                             paf.ctypes.data_as(C.POINTER(C.c_float)), paf.shape[0], paf.shape[1], paf.shape[2],
                                             int(img_height), conns.ctypes.data_as(C.POINTER(C.c_double)),
                                             counts.ctypes.data_as(C.POINTER(C.c_int)), special.ctypes.data_as(C.POINTER(C.c_int))),
               post.ctx)
    connected = [[] if special[k] else conns[k, :counts[k]].copy() for k in range(NUM_PAFS)]
    return connected, [int(k) for k in np.nonzero(special)[0]]


def find_humans(connected_limbs, special_limb, joint_list, test_cfg, joint2limb_pairs):
    """utils/parse_skeletons.py:413-600 on the GPU (k_assemble_py): -> (person_to_joint_assoc (P, 20, 2) float64,
    joint_candidates (N, 4) float64)."""
    _check_cfg(test_cfg)
    post = _ctx()
    jl = _joint_rows(joint_list)
    conns = np.zeros((NUM_PAFS, post.maxp, 6), np.float64)
    counts = np.zeros(NUM_PAFS, np.int32)
    for k in range(NUM_PAFS):
        if k in special_limb or len(connected_limbs[k]) == 0:
            continue
        c = np.asarray(connected_limbs[k], np.float64).reshape(-1, 6)
        counts[k] = len(c)
        conns[k, :len(c)] = c
    persons = np.zeros((128, 20, 2), np.float64)
    n = C.c_int(0)
    L = _lib.load()
    _lib.check(L.pp_py_find_humans_host(post.ctx, conns.ctypes.data_as(C.POINTER(C.c_double)),
                                        counts.ctypes.data_as(C.POINTER(C.c_int)), jl.ctypes.data_as(C.POINTER(C.c_float)), len(jl),
                                        persons.ctypes.data_as(C.POINTER(C.c_double)), 128, C.byref(n)), post.ctx)
    joint_candidates = np.array([item for sublist in joint_list for item in sublist], dtype=np.float64).reshape(-1, 4)
    return persons[: n.value].copy(), joint_candidates
