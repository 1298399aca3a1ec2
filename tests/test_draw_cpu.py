"""CPU: the NumPy renderer of the demo surface (utils/draw.py; demo_image.py:174-240, utils/common.py:240-264)."""
import numpy as np


def _human(parts):
    from utils.common import BodyPart, Human
    h = Human([])
    for k, (x, y) in parts.items():
        h.body_parts[k] = BodyPart("0-%d" % k, k, x, y, 0.9)
    h.score = 1.0
    return h


def test_joints_and_limbs_are_drawn_in_the_reference_colours():
    from utils import draw
    canvas = np.zeros((120, 160, 3), np.uint8)
    # neck (1), right shoulder (2), right elbow (3); nose (0) absent -> the (1, 0) pair is skipped
    out = draw.draw_humans(canvas, [_human({1: (80, 30), 2: (50, 40), 3: (40, 80)})], imgcopy=True)
    assert (canvas == 0).all()                                           # imgcopy leaves the input alone
    # pair (1, 2) is CocoPairsRender[0], pair (2, 3) is [2]: lines painted after the discs, so the line colour covers the centres
    assert tuple(out[40, 50]) == tuple(draw.CocoColors[2])               # end point of both lines: the later pair's colour
    assert tuple(out[80, 40]) == tuple(draw.CocoColors[2])
    assert tuple(out[35, 65]) == tuple(draw.CocoColors[0])               # midpoint of neck - right shoulder
    assert tuple(out[26, 80]) == tuple(draw.CocoColors[1])               # the neck disc (radius 4.5) above the line
    assert tuple(out[60, 45]) == tuple(draw.CocoColors[2])               # midpoint of shoulder - elbow
    assert (out[100:, :] == 0).all() and (out[:, 120:] == 0).all()       # nothing elsewhere
    # normalised coordinates (utils/common.py:252)
    out2 = draw.draw_humans(canvas, [_human({1: (0.5, 0.25), 2: (0.3125, 1 / 3)})], imgcopy=True, normalized=True)
    assert tuple(out2[35, 65]) == tuple(draw.CocoColors[0])


def test_original_drawing_blends_rotated_ellipses():
    from posepaf import skeleton as sk
    from utils import draw
    canvas = np.full((100, 100, 3), 100, np.uint8)
    persons = np.full((1, 20, 2), -1.0)
    persons[0, 1, 0], persons[0, 0, 0] = 0, 1                            # limb 0 = (neck 1, nose 0): peak ids 0 and 1
    cand = np.array([[20.0, 50.0, 0.9, 0], [80.0, 50.0, 0.9, 1]])
    out = draw.draw_limbs_original(canvas, persons, cand, sk.LIMB_PAIRS, sk.DRAW_LIST)
    want = np.rint(100 * 0.4 + np.array(draw.LimbColors[0]) * 0.6)       # addWeighted(canvas, 0.4, cur, 0.6)
    assert tuple(out[50, 50]) == tuple(want.astype(np.uint8))            # centre of the ellipse
    assert tuple(out[50 + 3, 50]) == tuple(want.astype(np.uint8)) and tuple(out[50 + 5, 50]) == (100, 100, 100)   # semi-minor axis 3
    assert tuple(out[50, 16]) == (40, 40, 40)                            # black ring around a joint, blended: 100*0.4 + 0*0.6
    assert sk.DRAW_LIST == [0] + list(range(5, 21)) + [29]


def test_clipping_at_the_image_border():
    from utils import draw
    canvas = np.zeros((20, 20, 3), np.uint8)
    draw.disc(canvas, (-2, -2), 4.5, [1, 2, 3])
    draw.line(canvas, (15, 10), (40, 10), [9, 9, 9], 3)
    draw.fill_ellipse(canvas, (10, 25), (8, 3), 0, [5, 5, 5])
    assert tuple(canvas[0, 0]) == (1, 2, 3) and tuple(canvas[10, 19]) == (9, 9, 9) and (canvas[5, 5] == 0).all()
