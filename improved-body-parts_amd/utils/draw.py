"""Skeleton rendering for the demo (demo_image.py:174-240, utils/common.py:240-264) without OpenCV: filled discs, thick
lines, rotated filled ellipses and alpha blending rasterised with NumPy on a BGR uint8 canvas.  The colour tables and the
pair list are the reference's data (utils/common.py:281-289, demo_image.py:30-33).  Pixel-exact parity with cv2's
rasteriser is not claimed (cv2 is absent here: unpinned); geometry, colours, draw order and blending weights are."""
import math

import numpy as np

# utils/common.py:281-283: one colour per joint / per rendered pair (B, G, R as the reference passes them to cv2)
CocoColors = [[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0],
              [0, 255, 85], [0, 255, 170], [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255],
              [170, 0, 255], [255, 0, 255], [255, 0, 170], [255, 0, 85]]
# utils/common.py:285-289
CocoPairs = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11), (11, 12), (12, 13), (1, 0),
             (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]
CocoPairsRender = CocoPairs[:-2]
# demo_image.py:30-33: limb colours of the original (non-refactored) drawing
LimbColors = [[128, 114, 250], [130, 238, 238], [48, 167, 238], [180, 105, 255], [255, 0, 0], [255, 85, 0], [255, 170, 0],
              [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0], [0, 255, 85], [0, 255, 170], [0, 255, 255],
              [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255], [170, 0, 255], [255, 0, 255], [255, 0, 170],
              [255, 0, 85], [193, 193, 255], [106, 106, 255], [20, 147, 255]]


def _box(canvas, x0, y0, x1, y1):
    h, w = canvas.shape[:2]
    return max(0, int(x0)), max(0, int(y0)), min(w, int(x1) + 1), min(h, int(y1) + 1)


def disc(canvas, center, radius, color):
    """filled disc (cv2.circle with thickness 3 on radius 3 fills it: outer radius 4.5)"""
    cx, cy = center
    x0, y0, x1, y1 = _box(canvas, cx - radius - 1, cy - radius - 1, cx + radius + 1, cy + radius + 1)
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    canvas[y0:y1, x0:x1][(xx - cx) ** 2 + (yy - cy) ** 2 <= radius * radius] = color


def ring(canvas, center, radius, color, thickness=2):
    cx, cy = center
    r1 = radius + thickness / 2.0
    x0, y0, x1, y1 = _box(canvas, cx - r1 - 1, cy - r1 - 1, cx + r1 + 1, cy + r1 + 1)
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    d2 = (xx - cx) ** 2 + (yy - cy) ** 2
    canvas[y0:y1, x0:x1][(d2 <= r1 * r1) & (d2 >= (radius - thickness / 2.0) ** 2)] = color


def line(canvas, p0, p1, color, thickness=3):
    """all pixels whose centre is within thickness / 2 of the segment"""
    (ax, ay), (bx, by) = p0, p1
    t = thickness / 2.0
    x0, y0, x1, y1 = _box(canvas, min(ax, bx) - t - 1, min(ay, by) - t - 1, max(ax, bx) + t + 1, max(ay, by) + t + 1)
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float64)
    dx, dy = bx - ax, by - ay
    den = float(dx * dx + dy * dy)
    u = np.clip(((xx - ax) * dx + (yy - ay) * dy) / den, 0.0, 1.0) if den > 0 else np.zeros_like(xx)
    d2 = (xx - (ax + u * dx)) ** 2 + (yy - (ay + u * dy)) ** 2
    canvas[y0:y1, x0:x1][d2 <= t * t] = color


def fill_ellipse(canvas, center, axes, angle_deg, color):
    """filled ellipse with semi-axes `axes` rotated by angle_deg (cv2.ellipse2Poly + fillConvexPoly, demo_image.py:232-236)"""
    cx, cy = center
    a, b = max(float(axes[0]), 0.5), max(float(axes[1]), 0.5)
    r = max(a, b) + 1
    x0, y0, x1, y1 = _box(canvas, cx - r, cy - r, cx + r, cy + r)
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float64)
    c, s = math.cos(math.radians(angle_deg)), math.sin(math.radians(angle_deg))
    u = (xx - cx) * c + (yy - cy) * s
    v = -(xx - cx) * s + (yy - cy) * c
    canvas[y0:y1, x0:x1][(u / a) ** 2 + (v / b) ** 2 <= 1.0] = color


def draw_humans(npimg, humans, imgcopy=False, normalized=False):
    """utils/common.py:240-264 (normalized=True: coordinates in [0, 1] as there) and the refactored branch of
    demo_image.py:174-192 (normalized=False: pixel coordinates): joints as discs, CocoPairsRender as lines."""
    if imgcopy:
        npimg = np.copy(npimg)
    image_h, image_w = npimg.shape[:2]
    for human in humans:
        centers = {}
        for i in range(18):
            if i not in human.body_parts:
                continue
            bp = human.body_parts[i]
            centers[i] = (int(bp.x * image_w + 0.5), int(bp.y * image_h + 0.5)) if normalized else (int(bp.x), int(bp.y))
            disc(npimg, centers[i], 4.5, CocoColors[i])
        for pair_order, pair in enumerate(CocoPairsRender):
            if pair[0] in centers and pair[1] in centers:
                line(npimg, centers[pair[0]], centers[pair[1]], CocoColors[pair_order], 3)
    return npimg


def draw_limbs_original(canvas, person_to_joint_assoc, joint_candidates, joint2limb_pairs, draw_list):
    """the non-refactored drawing of demo_image.py:218-240: per limb type and person a rotated filled ellipse between the two
    joints, blended 0.4 / 0.6 with the canvas so far; black rings on the joints"""
    color_board = [0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21]
    for color_idx, i in enumerate(draw_list):
        for person in person_to_joint_assoc:
            ids = person[np.asarray(joint2limb_pairs[i]), 0].astype(int)
            if -1 in ids:
                continue
            cur = canvas.copy()
            xs, ys = joint_candidates[ids, 0], joint_candidates[ids, 1]
            length = float(((ys[0] - ys[1]) ** 2 + (xs[0] - xs[1]) ** 2) ** 0.5)
            angle = math.degrees(math.atan2(ys[0] - ys[1], xs[0] - xs[1]))
            ring(cur, (int(xs[0]), int(ys[0])), 4, [0, 0, 0], 2)
            ring(cur, (int(xs[1]), int(ys[1])), 4, [0, 0, 0], 2)
            fill_ellipse(cur, (int(np.mean(xs)), int(np.mean(ys))), (int(length / 2), 3), int(angle), LimbColors[color_board[color_idx % 18]])
            canvas = np.clip(np.rint(canvas.astype(np.float64) * 0.4 + cur.astype(np.float64) * 0.6), 0, 255).astype(np.uint8)
    return canvas
