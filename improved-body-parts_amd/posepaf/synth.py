"""Synthetic network-output generator (benchmark / test DATA, not an oracle).

There are no pretrained weights and no COCO images offline, and a randomly initialised IMHN
produces no peaks, so post-processing load is injected from ground-truth-style maps that follow
the recipe the reference trains against (py_cocodata_server/py_data_heatmapper.py:105-257):

* keypoint channel k: max over people of exp(-d^2 / (2*9^2)) evaluated at the stride-4 grid
  centres 4*i + 1.5, inside a +-7 cell window (:117-164, sigma 9, gaussian_size 14);
* limb channel l ("PAF" here is a SCALAR body-part map, one channel per limb): Gaussian
  (sigma 7) of the perpendicular distance to the segment inside the end-point bounding box
  grown by 4 px, responses <= 0.015 set to 0.01, averaged where limbs overlap (:178-244,
  :326-357);
* clip to [0, 1]; optional additive noise; optional binary16 round-trip (AMP output dtype,
  utils/parse_skeletons.py:75).

The second ("flipped") sample is the mirrored scene with left/right channels exchanged and its
own noise, so that the flip-average of utils/parse_skeletons.py:91-93 does real work.
"""
from __future__ import annotations

import numpy as np

from . import skeleton as sk

# a standing person in a unit-height box, OpenPose/CMU 18-part order (config/config.py part_str)
_TEMPLATE = np.array(
    [
        [0.50, 0.10], [0.50, 0.20], [0.38, 0.21], [0.33, 0.36], [0.30, 0.50], [0.62, 0.21],
        [0.67, 0.36], [0.70, 0.50], [0.43, 0.52], [0.42, 0.72], [0.41, 0.92], [0.57, 0.52],
        [0.58, 0.72], [0.59, 0.92], [0.47, 0.08], [0.53, 0.08], [0.43, 0.09], [0.57, 0.09],
    ],
    dtype=np.float64,
)

SIGMA_KP = 9.0
SIGMA_LIMB = 7.0
KP_WINDOW = 7          # gaussian_size // 2 with gaussian_size = 14
LIMB_GROW = 4.0        # paf_thre = 1 * stride
LIMB_FLOOR_THRE = 0.015
LIMB_FLOOR_VALUE = 0.01
STRIDE = 4


def random_people(n_people: int, rng: np.random.Generator, img_h: int = 512, img_w: int = 512,
                  p_missing: float = 0.08) -> np.ndarray:
    """Returns joints (P, 18, 3): x, y in image pixels, flag 1 = present, 2 = absent."""
    joints = np.zeros((n_people, sk.NUM_PART, 3), dtype=np.float64)
    for p in range(n_people):
        height = rng.uniform(0.22, 0.62) * img_h
        cx = rng.uniform(0.08, 0.92) * img_w
        cy = rng.uniform(0.05, 0.95) * img_h
        pts = (_TEMPLATE - np.array([0.5, 0.5])) * height
        pts += rng.normal(0.0, 0.025 * height, size=pts.shape)
        ang = rng.normal(0.0, 0.15)
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        pts = pts @ rot.T + np.array([cx, cy])
        joints[p, :, :2] = pts
        present = rng.random(sk.NUM_PART) >= p_missing
        inside = (pts[:, 0] >= 0) & (pts[:, 0] < img_w) & (pts[:, 1] >= 0) & (pts[:, 1] < img_h)
        joints[p, :, 2] = np.where(present & inside, 1.0, 2.0)
    return joints


def render_maps(joints: np.ndarray, h: int, w: int) -> np.ndarray:
    """GT-style (50, h, w) float32 planar maps for `joints` (P,18,3) on a stride-4 grid."""
    maps = np.zeros((sk.NUM_CH, h, w), dtype=np.float32)
    gx = np.arange(w, dtype=np.float32) * STRIDE + STRIDE / 2 - 0.5
    gy = np.arange(h, dtype=np.float32) * STRIDE + STRIDE / 2 - 0.5
    two_s2 = np.float32(2 * SIGMA_KP * SIGMA_KP)
    for part in range(sk.NUM_PART):
        ch = maps[sk.NUM_LIMB + part]
        for p in range(joints.shape[0]):
            if joints[p, part, 2] >= 2:
                continue
            x, y = np.float32(joints[p, part, 0]), np.float32(joints[p, part, 1])
            cx, cy = int(round(float(x) / STRIDE)), int(round(float(y) / STRIDE))
            x0, x1 = max(cx - KP_WINDOW, 0), min(cx + KP_WINDOW + 1, w)
            y0, y1 = max(cy - KP_WINDOW, 0), min(cy + KP_WINDOW + 1, h)
            if x1 <= x0 or y1 <= y0:
                continue
            ex = np.exp(-((gx[x0:x1] - x) ** 2) / two_s2)
            ey = np.exp(-((gy[y0:y1] - y) ** 2) / two_s2)
            np.maximum(ch[y0:y1, x0:x1], np.outer(ey, ex), out=ch[y0:y1, x0:x1])
    X, Y = np.meshgrid(gx, gy)
    for limb, (a, b) in enumerate(sk.LIMB_PAIRS):
        acc = np.zeros((h, w), dtype=np.float32)
        cnt = np.zeros((h, w), dtype=np.float32)
        for p in range(joints.shape[0]):
            if joints[p, a, 2] >= 2 or joints[p, b, 2] >= 2:
                continue
            x1_, y1_ = joints[p, a, :2]
            x2_, y2_ = joints[p, b, :2]
            dx, dy = x2_ - x1_, y2_ - y1_
            norm = float(np.hypot(dx, dy))
            if norm == 0.0:
                continue
            sx0 = max(int(round((min(x1_, x2_) - LIMB_GROW) / STRIDE)), 0)
            sy0 = max(int(round((min(y1_, y2_) - LIMB_GROW) / STRIDE)), 0)
            sx1 = int(round((max(x1_, x2_) + LIMB_GROW) / STRIDE))
            sy1 = int(round((max(y1_, y2_) + LIMB_GROW) / STRIDE))
            if sx1 < 0 or sy1 < 0:
                continue
            xs, ys = slice(sx0, sx1 + 1), slice(sy0, sy1 + 1)
            d = np.abs(dx * (y1_ - Y[ys, xs]) - (x1_ - X[ys, xs]) * dy) / (norm + 1e-6)
            g = np.exp(-(d ** 2) / (2 * SIGMA_LIMB ** 2)).astype(np.float32)
            g[g <= LIMB_FLOOR_THRE] = LIMB_FLOOR_VALUE
            acc[ys, xs] += g
            cnt[ys, xs] += 1
        np.divide(acc, cnt, out=acc, where=cnt > 0)
        maps[limb] = acc
    np.clip(maps, 0.0, 1.0, out=maps)
    return maps


def mirror_sample(maps: np.ndarray) -> np.ndarray:
    """What the network would output for the mirrored image: out1[k] = mirror_W(out0[flip_ord[k]]).

    flip_paf_ord / flip_heat_ord are involutions (config/config.py:150-152), so applying the
    reference's un-flip (utils/parse_skeletons.py:91-93) to this returns the original scene."""
    out = np.empty_like(maps)
    out[: sk.NUM_LIMB] = maps[: sk.NUM_LIMB][sk.FLIP_PAF_ORD][:, :, ::-1]
    out[sk.NUM_LIMB:] = maps[sk.NUM_LIMB:][sk.FLIP_HEAT_ORD][:, :, ::-1]
    return np.ascontiguousarray(out)


def make_net_output(n_people: int, seed: int, h: int = 128, w: int = 128, noise: float = 0.02,
                    dtype=np.float16, flip: bool = True, p_missing: float = 0.08) -> np.ndarray:
    """One image's network output, shape (2, 50, h, w) (or (1,50,h,w) without flip), `dtype`.

    Deterministic in (n_people, seed, h, w, noise, dtype)."""
    return make_scene(n_people, seed, h, w, noise, dtype, flip, p_missing)[0]


def make_scene(n_people: int, seed: int, h: int = 128, w: int = 128, noise: float = 0.02,
               dtype=np.float16, flip: bool = True, p_missing: float = 0.08):
    """-> (network output as make_net_output, ground-truth joints (P, 18, 3) in image pixels)."""
    rng = np.random.default_rng(seed)
    joints = random_people(n_people, rng, img_h=h * STRIDE, img_w=w * STRIDE, p_missing=p_missing)
    base = render_maps(joints, h, w)
    samples = [base]
    if flip:
        samples.append(mirror_sample(base))
    out = np.stack(samples).astype(np.float32)
    if noise > 0:
        out = out + rng.normal(0.0, noise, size=out.shape).astype(np.float32)
    return np.ascontiguousarray(out.astype(dtype)), joints


def make_scene_at_scales(n_people: int, seed: int, sizes, noise: float = 0.02, dtype=np.float16, p_missing: float = 0.08,
                         img: int = 512):
    """The SAME people rendered on feature maps of several sizes (a scale search): -> (list of (2,50,h,w) outputs, joints).
    sizes: list of (h, w, scale) with the map showing the image scaled by `scale` (top-left aligned, padding beyond)."""
    rng = np.random.default_rng(seed)
    joints = random_people(n_people, rng, img_h=img, img_w=img, p_missing=p_missing)
    outs = []
    for (h, w, scale) in sizes:
        j = joints.copy()
        j[:, :, :2] *= scale
        base = render_maps(j, h, w)
        out = np.stack([base, mirror_sample(base)]).astype(np.float32)
        if noise > 0:
            out = out + rng.normal(0.0, noise, size=out.shape).astype(np.float32)
        outs.append(np.ascontiguousarray(out.astype(dtype)))
    return outs, joints


def make_id_sum_merge_scene(weak: float = 0.8) -> np.ndarray:
    """A hand-built (1, 50, 128, 128) float32 output (flip off) that drives pafprocess.cpp's merge (:222-228) into
    ADDING two real peak ids: skeleton {neck 3, nose 1, Rsho 4} absorbs skeleton {nose 0, Reye 6} through the
    connection nose 0 - Rsho 4 of limb (0, 2); `id > 0` membership (:200-201) does not see nose id 0, so the nose column
    becomes 1 + 0 + 1 = 2 -- the id of the third nose.  The weaker connection nose 2 - Rsho 5 that follows in the same limb
    then matches the merged skeleton and is dropped instead of founding a person: the reference reports ONE person with
    nose id 2.  Each "person" rendered here is just the two end points of one wanted connection."""
    def two(parts):
        j = np.zeros((sk.NUM_PART, 3))
        j[:, 2] = 2
        for p, (x, y) in parts.items():
            j[p] = (x, y, 1)
        return j
    n0, n1, n2 = (80, 80), (240, 80), (400, 80)
    k1, b, c, e0 = (240, 160), (160, 200), (400, 200), (80, 30)
    pieces = [two({1: k1, 0: n1}), two({0: n0, 14: e0}), two({1: k1, 2: b}), two({0: n0, 2: b}), two({0: n2, 2: c})]
    maps = render_maps(np.stack(pieces), 128, 128)
    limb = [i for i, pr in enumerate(map(tuple, sk.LIMB_PAIRS)) if pr == (0, 2)][0]
    maps[limb, :, 80:] *= np.float32(weak)  # nose 2 - Rsho 5 ranks after nose 0 - Rsho 4
    return np.ascontiguousarray(maps[None].astype(np.float32))
