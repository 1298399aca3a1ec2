"""`Human` / `BodyPart` result containers with the reference's attribute names (utils/common.py:39-51, :267-279).
Drawing and box helpers of that file are visualisation and out of scope."""
from enum import Enum


class CocoPart(Enum):
    Nose = 0
    Neck = 1
    RShoulder = 2
    RElbow = 3
    RWrist = 4
    LShoulder = 5
    LElbow = 6
    LWrist = 7
    RHip = 8
    RKnee = 9
    RAnkle = 10
    LHip = 11
    LKnee = 12
    LAnkle = 13
    REye = 14
    LEye = 15
    REar = 16
    LEar = 17
    Background = 18


class BodyPart:
    __slots__ = ("uidx", "part_idx", "x", "y", "score")

    def __init__(self, uidx, part_idx, x, y, score):
        self.uidx = uidx
        self.part_idx = part_idx
        self.x, self.y = x, y
        self.score = score

    def get_part_name(self):
        return CocoPart(self.part_idx)

    def __str__(self):
        return "BodyPart:%d-(%.2f, %.2f) score=%.2f" % (self.part_idx, self.x, self.y, self.score)

    __repr__ = __str__


class Human:
    __slots__ = ("body_parts", "pairs", "uidx_list", "score")

    def __init__(self, pairs):
        self.pairs = []
        self.uidx_list = set()
        self.body_parts = {}
        for pair in pairs:
            self.add_pair(pair)
        self.score = 0.0

    @staticmethod
    def _get_uidx(part_idx, idx):
        return "%d-%d" % (part_idx, idx)

    def add_pair(self, pair):
        self.pairs.append(pair)
        for part_idx, idx, coord, score in ((pair.part_idx1, pair.idx1, pair.coord1, pair.score),
                                            (pair.part_idx2, pair.idx2, pair.coord2, pair.score)):
            uid = Human._get_uidx(part_idx, idx)
            self.body_parts[part_idx] = BodyPart(uid, part_idx, coord[0], coord[1], score)
            self.uidx_list.add(uid)

    def part_count(self):
        return len(self.body_parts.keys())

    def get_max_score(self):
        return max(x.score for _, x in self.body_parts.items())

    def __str__(self):
        return " ".join(str(x) for x in self.body_parts.values())

    __repr__ = __str__
