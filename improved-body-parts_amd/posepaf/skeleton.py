"""Skeleton definition and post-processing constants of the reference, restated as plain data.

Sources (relative to /root/reference): 18 parts + 30 limbs config/config.py:60-121 and
utils/pafprocess/pafprocess.h:20-27; flip permutations config/config.py:150-152; channel layout
config/config.py:125-132; C++ thresholds utils/pafprocess/pafprocess.h:6-18; INI thresholds
utils/config:12-27; COCO keypoint order evaluate.py:40.

tests/test_constants.py asserts these against values captured from the imported reference
(tests/golden/constants.json).
"""
import numpy as np

# config/config.py:60-62 (the INI's part_str, utils/config:40, lists eyes/ears in the opposite order; the model
# and the flip permutations follow config.py)
PARTS = ["nose", "neck", "Rsho", "Relb", "Rwri", "Lsho", "Lelb", "Lwri", "Rhip", "Rkne", "Rank",
         "Lhip", "Lkne", "Lank", "Reye", "Leye", "Rear", "Lear"]
PART_STR_INI = ["nose", "neck", "Rsho", "Relb", "Rwri", "Lsho", "Lelb", "Lwri", "Rhip", "Rkne", "Rank",
                "Lhip", "Lkne", "Lank", "Leye", "Reye", "Lear", "Rear", "pt19"]
NUM_PART = 18
NUM_LIMB = 30
NUM_HEAT = NUM_PART + 2          # 18 keypoint maps + 2 background maps
NUM_CH = NUM_LIMB + NUM_HEAT     # 50; [0:30] limb maps, [30:48] keypoints, [48:50] background
STRIDE = 4
MAX_DOWNSAMPLE = 64
PAD_VALUE = 128
BOXSIZE = 512

LIMB_FROM = [1, 1, 1, 1, 1, 0, 0, 14, 15, 1, 2, 3, 1, 5, 6, 1, 8, 9, 1, 11, 12, 0, 0, 2, 8, 5, 11, 16, 17, 8]
LIMB_TO = [0, 14, 15, 16, 17, 14, 15, 16, 17, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 2, 5, 8, 12, 11, 9, 2, 5, 11]
LIMB_PAIRS = list(zip(LIMB_FROM, LIMB_TO))

FLIP_HEAT_ORD = np.array([0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16, 18, 19])
FLIP_PAF_ORD = np.array([0, 2, 1, 4, 3, 6, 5, 8, 7, 12, 13, 14, 9, 10, 11, 18, 19, 20, 15, 16, 17, 22, 21, 25, 26,
                         23, 24, 28, 27, 29])

# CMU 18-part order -> COCO 17-keypoint order (evaluate.py:40)
ORDER_COCO = [0, 15, 14, 17, 16, 5, 2, 6, 3, 7, 4, 11, 8, 12, 9, 13, 10]
DT_GT_MAPPING = {0: 0, 1: None, 2: 6, 3: 8, 4: 10, 5: 5, 6: 7, 7: 9, 8: 12, 9: 14, 10: 16, 11: 11, 12: 13,
                 13: 15, 14: 2, 15: 1, 16: 4, 17: 3}

# utils/pafprocess/pafprocess.h:6-18 (the C++ path ignores the INI file)
THRESH_HEAT = 0.05
THRESH_PAF_SCORE = 0.1
THRESH_PAF_STEP_RATIO = 0.8
THRESH_PART_CNT = 2
THRESH_SKELETON_SCORE = 0.45
STEP_PAF = 20
LIMB_LENGTH_RATE = 16
MIN_SCORE_TOLERANCE = 0.7
PAF_OUT_WEIGHTS = (0.5, 0.25, 0.25)
NMS_THRESHOLD = 0.1              # hard-coded in heatmap_nms, utils/parse_skeletons.py:139
NMS_WIN_SIZE = 2                 # utils/parse_skeletons.py:135


def default_test_cfg():
    """`param` dict of utils/config_reader.py:6-37 (utils/config [param] section), typed."""
    return {
        "use_gpu": 1, "GPUdeviceNumber": 0, "modelID": "1", "starting_range": 0.8, "ending_range": 2.0,
        "scale_search": [0.5, 1.0, 1.5, 2.0, 3.0], "rotation_search": [0.0], "thre1": 0.1, "thre2": 0.1,
        "connect_ration": 0.8, "min_num": 4, "mid_num": 20, "len_rate": 16.0, "connection_tole": 0.7,
        "crop_ratio": 2.5, "bbox_ratio": 0.25, "offset_radius": 2, "remove_recon": 0,
    }


def default_model_cfg():
    """`model` dict of utils/config_reader.py (utils/config [[1]] section), typed."""
    return {"boxsize": BOXSIZE, "padValue": PAD_VALUE, "np": "12", "stride": STRIDE,
            "max_downsample": MAX_DOWNSAMPLE, "part_str": list(PART_STR_INI)}

# config/config.py:154: the limb types the original demo drawing renders
DRAW_LIST = [0] + list(range(5, 21)) + [29]
