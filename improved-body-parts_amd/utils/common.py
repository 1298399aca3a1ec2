"""Result containers of the inference path (A9), with the attribute names evaluate.py:111-129 and :182-209 read:
`human.body_parts[part_idx] -> BodyPart(.x, .y, .score, .part_idx, .uidx)` and `human.score`.

Only what posepaf/coco.py fills and the COCO writer reads is kept.  The reference's pair bookkeeping (Human.add_pair,
uidx_list, part_count, get_max_score; utils/common.py:267-300) belongs to a grouping scheme this path never runs, and its
drawing / box helpers are visualisation: both are out of scope."""
from enum import IntEnum

# part index -> name, in the order of the network's keypoint channels (config/config.py:96-104)
CocoPart = IntEnum("CocoPart", ["Nose", "Neck", "RShoulder", "RElbow", "RWrist", "LShoulder", "LElbow", "LWrist", "RHip",
                                "RKnee", "RAnkle", "LHip", "LKnee", "LAnkle", "REye", "LEye", "REar", "LEar", "Background"],
                   start=0)


class BodyPart:
    """One detected joint of one person: image coordinates, peak score, part index, and a printable id."""
    __slots__ = ("uidx", "part_idx", "x", "y", "score")

    def __init__(self, uidx, part_idx, x, y, score):
        self.uidx, self.part_idx, self.x, self.y, self.score = uidx, part_idx, x, y, score

    def get_part_name(self):
        return CocoPart(self.part_idx)

    def __repr__(self):
        return f"BodyPart({self.get_part_name().name} @ ({self.x}, {self.y}), score {self.score:.3f})"


class Human:
    """One assembled person: `body_parts` maps part index -> BodyPart; `score` is pafprocess's get_score()."""
    __slots__ = ("body_parts", "score")

    def __init__(self, pairs=()):
        if len(pairs):
            raise NotImplementedError("pair-based construction (utils/common.py:267-290) is not part of the inference path")
        self.body_parts = {}
        self.score = 0.0

    def __repr__(self):
        return f"Human(score {self.score:.3f}, parts {sorted(self.body_parts)})"
