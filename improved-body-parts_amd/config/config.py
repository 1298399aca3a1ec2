"""`TrainingOpt` / `GetConfig("Canonical")` with the fields the inference path reads
(reference config/config.py:8-22, :57-152, :270-287).  Training hyper-parameters are out of scope."""
import numpy as np

from posepaf import skeleton as sk


class TrainingOpt:
    config_name = "Canonical"
    nstack = 4
    hourglass_inp_dim = 256
    increase = 128
    ckpt_path = "./weights/simplepose_imhn_epoch52.pth"


class CanonicalConfig:
    def __init__(self):
        self.width = 512
        self.height = 512
        self.stride = sk.STRIDE
        self.parts = list(sk.PARTS)
        self.num_parts = len(self.parts)
        self.parts_dict = dict(zip(self.parts, range(self.num_parts)))
        self.parts += ["background", "reverseKeypoint"]  # config/config.py:67-69
        self.num_parts_with_background = len(self.parts)
        self.limb_from = list(sk.LIMB_FROM)
        self.limb_to = list(sk.LIMB_TO)
        self.limbs_conn = np.array(sk.LIMB_PAIRS)
        self.paf_layers = len(self.limbs_conn)
        self.heat_layers = self.num_parts
        self.num_layers = self.paf_layers + self.heat_layers + 2
        self.paf_start = 0
        self.heat_start = self.paf_layers
        self.bkg_start = self.paf_layers + self.heat_layers
        self.dt_gt_mapping = dict(sk.DT_GT_MAPPING)
        self.flip_heat_ord = sk.FLIP_HEAT_ORD.copy()
        self.flip_paf_ord = sk.FLIP_PAF_ORD.copy()


Configs = {"Canonical": CanonicalConfig}


def GetConfig(config_name):
    return Configs[config_name]()
