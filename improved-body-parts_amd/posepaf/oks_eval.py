"""COCO keypoint AP (OKS) evaluation without pycocotools (absent offline; evaluate.py:274-279 calls COCOeval).

Restates the published COCOeval procedure for iouType='keypoints', one category, area range 'all', maxDets 20:
OKS between a detection and a ground truth = mean over labelled GT keypoints of exp(-d^2 / (2 * area * (2*sigma_k)^2)),
detections sorted by score are greedily matched per image at each OKS threshold 0.50:0.05:0.95 (a detection takes the
unmatched GT with the highest OKS >= threshold), precision is made monotone and sampled at 101 recall points.
Parity against pycocotools is UNPINNED (library absent, no fixture in the reference); tests check hand-computed cases.
"""
from __future__ import annotations

import numpy as np

# COCO keypoint sigmas (17 keypoints, COCO order) as published with the COCO API
KPT_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
OKS_THRESHOLDS = np.linspace(0.5, 0.95, 10)
RECALL_POINTS = np.linspace(0.0, 1.0, 101)
MAX_DETS = 20


def compute_oks(dt_kpts: np.ndarray, gt_kpts: np.ndarray, gt_area: float, gt_bbox=None) -> float:
    """dt_kpts, gt_kpts: (17, 3) [x, y, v]."""
    vis = gt_kpts[:, 2] > 0
    k1 = int(vis.sum())
    var = (KPT_SIGMAS * 2) ** 2
    dx = dt_kpts[:, 0] - gt_kpts[:, 0]
    dy = dt_kpts[:, 1] - gt_kpts[:, 1]
    if k1 == 0:
        if gt_bbox is None:
            return 0.0
        x0, y0, w, h = gt_bbox  # no labelled keypoints: distance to the doubled box, as COCOeval does
        x1, y1 = x0 + 2 * w, y0 + 2 * h
        x0, y0 = x0 - w, y0 - h
        dx = np.maximum(0, x0 - dt_kpts[:, 0]) + np.maximum(0, dt_kpts[:, 0] - x1)
        dy = np.maximum(0, y0 - dt_kpts[:, 1]) + np.maximum(0, dt_kpts[:, 1] - y1)
    e = (dx ** 2 + dy ** 2) / var / (gt_area + np.spacing(1)) / 2
    if k1 > 0:
        e = e[vis]
    return float(np.sum(np.exp(-e)) / e.shape[0])


def evaluate_keypoints(gts: dict, dts: dict) -> dict:
    """gts: image_id -> list of {"keypoints": (51,), "area": float, "bbox": [x,y,w,h], "num_keypoints": int, "iscrowd": 0}
    dts: image_id -> list of {"keypoints": (51,), "score": float}.  Returns {"AP", "AP50", "AP75", "AR"}."""
    T = len(OKS_THRESHOLDS)
    dt_scores, dt_match = [], [[] for _ in range(T)]
    n_gt = 0
    for img_id in sorted(set(gts) | set(dts)):
        g = [x for x in gts.get(img_id, [])]
        d = sorted(dts.get(img_id, []), key=lambda r: -r["score"])[:MAX_DETS]
        ignore = np.array([bool(x.get("iscrowd", 0)) or x.get("num_keypoints", 1) == 0 for x in g], bool)
        n_gt += int((~ignore).sum())
        order = np.argsort(ignore, kind="mergesort")  # non-ignored GT first
        g = [g[i] for i in order]
        ignore = ignore[order]
        crowd = np.array([bool(x.get("iscrowd", 0)) for x in g], bool)
        oks = np.zeros((len(d), len(g)))
        for i, dd in enumerate(d):
            for j, gg in enumerate(g):
                oks[i, j] = compute_oks(np.asarray(dd["keypoints"], float).reshape(17, 3),
                                        np.asarray(gg["keypoints"], float).reshape(17, 3), float(gg["area"]), gg.get("bbox"))
        for ti, thr in enumerate(OKS_THRESHOLDS):
            gt_taken = np.zeros(len(g), bool)
            for i in range(len(d)):
                best, best_j = min(thr, 1 - 1e-10), -1
                for j in range(len(g)):
                    if gt_taken[j] and not crowd[j]:   # a crowd region may absorb any number of detections
                        continue
                    if best_j > -1 and not ignore[best_j] and ignore[j]:
                        break
                    if oks[i, j] < best:
                        continue
                    best, best_j = oks[i, j], j
                if best_j >= 0:
                    gt_taken[best_j] = True
                    dt_match[ti].append(-1 if ignore[best_j] else 1)  # -1: matched an ignored GT -> not counted
                else:
                    dt_match[ti].append(0)
        dt_scores.extend(r["score"] for r in d)
    scores = np.asarray(dt_scores, float)
    order = np.argsort(-scores, kind="mergesort")
    precisions = np.zeros((T, len(RECALL_POINTS)))
    recalls = np.zeros(T)
    for ti in range(T):
        m = np.asarray(dt_match[ti], int)[order] if len(scores) else np.zeros(0, int)
        tp = np.cumsum(m == 1).astype(float)
        fp = np.cumsum(m == 0).astype(float)
        if n_gt == 0 or len(m) == 0:
            continue
        rc = tp / n_gt
        pr = tp / (tp + fp + np.spacing(1))
        recalls[ti] = rc[-1]
        for i in range(len(pr) - 1, 0, -1):
            if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
        inds = np.searchsorted(rc, RECALL_POINTS, side="left")
        q = np.zeros(len(RECALL_POINTS))
        valid = inds < len(pr)
        q[valid] = pr[inds[valid]]
        precisions[ti] = q
    ap = float(precisions.mean()) if n_gt else float("nan")
    return {"AP": ap, "AP50": float(precisions[0].mean()), "AP75": float(precisions[5].mean()),
            "AR": float(recalls.mean()), "n_gt": n_gt, "n_dt": int(len(scores))}


def gt_from_synth_joints(joints: np.ndarray) -> list:
    """synth.random_people joints (P, 18, 3; CMU order, flag 1 = present) -> COCO-style GT annotations."""
    from . import skeleton as sk
    out = []
    for p in range(joints.shape[0]):
        kp = np.zeros((17, 3))
        for coco_i, cmu_i in enumerate(sk.ORDER_COCO):
            if joints[p, cmu_i, 2] < 2:
                kp[coco_i] = (joints[p, cmu_i, 0], joints[p, cmu_i, 1], 2)
        vis = kp[:, 2] > 0
        if vis.sum() == 0:
            continue
        x0, y0 = kp[vis, 0].min(), kp[vis, 1].min()
        x1, y1 = kp[vis, 0].max(), kp[vis, 1].max()
        w, h = max(x1 - x0, 1.0), max(y1 - y0, 1.0)
        out.append({"keypoints": kp.reshape(-1).tolist(), "area": float(w * h), "bbox": [float(x0), float(y0), float(w), float(h)],
                    "num_keypoints": int(vis.sum()), "iscrowd": 0})
    return out
