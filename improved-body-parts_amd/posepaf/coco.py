"""Result containers and COCO formatting (A9): utils/common.py:39-51,:267-279 and evaluate.py:97-129,:182-209."""
from __future__ import annotations

import numpy as np

from . import skeleton as sk
from utils.common import BodyPart, Human


def humans_from_record(rec) -> list:
    """evaluate.py:111-129: one Human per skeleton that has at least one part."""
    humans = []
    for h in range(int(rec["n_humans"])):
        row = rec["humans"][h]
        human = Human([])
        added = False
        for part in range(sk.NUM_PART):
            pid = int(row["peak_id"][part])
            if pid < 0:
                continue
            added = True
            human.body_parts[part] = BodyPart("%d-%d" % (h, part), part, int(row["x"][part]), int(row["y"][part]),
                                              float(row["part_score"][part]))
        if added:
            human.score = float(row["score"])
            humans.append(human)
    return humans


def coco_results(image_id, humans) -> list:
    """evaluate.py:182-209 (`append_result`, refactored branch): 17 COCO keypoints (x, y, 1) in ORDER_COCO, zeros
    for missing parts, score = human.score."""
    out = []
    for human in humans:
        kp = np.zeros((sk.NUM_PART, 3), dtype=np.float64)
        for i in range(sk.NUM_PART):
            if i in human.body_parts:
                bp = human.body_parts[i]
                kp[i] = (bp.x, bp.y, 1)
        coco = kp[sk.ORDER_COCO, :]
        out.append({"image_id": image_id, "category_id": 1, "keypoints": list(coco.reshape(17 * 3)), "score": human.score})
    return out
