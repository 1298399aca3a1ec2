"""ctypes bindings for the CHECKERS -- TEST INFRASTRUCTURE ONLY.

* `Oracle`    : our plain-C restatement (oracle/posepaf_oracle.c -> _build/libposepaf_oracle.so)
* `Reference` : the reference's own C++ `pafprocess` compiled as-is into _ref/libpafprocess_ref.so
                (mangled C++ symbols of utils/pafprocess/pafprocess.h:70-76)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does (tests/test_no_oracle_in_product.py enforces it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libposepaf_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libpafprocess_ref.so")

NUM_PART, NUM_LIMB, NUM_HEAT, NUM_CH = 18, 30, 20, 50


def build(force: bool = False) -> None:
    """Compile the checkers (gcc only; `make -C oracle`)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "posepaf_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "_build/libposepaf_oracle.so"], check=True, capture_output=True)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(REF_SO)):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


class _Conn(C.Structure):
    _fields_ = [("cid1", C.c_int), ("cid2", C.c_int), ("score", C.c_float), ("peak_id1", C.c_int),
                ("peak_id2", C.c_int), ("length", C.c_float)]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Oracle:
    def __init__(self):
        build()
        L = C.CDLL(ORACLE_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_process_paf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_float), C.c_int]
        for name, res in [("orc_get_num_humans", C.c_int), ("orc_get_num_peaks", C.c_int), ("orc_get_sort_oob", C.c_int)]:
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = res
        L.orc_get_part_peak_id.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_get_score.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_score.restype = C.c_float
        for name, res in [("orc_get_part_x", C.c_int), ("orc_get_part_y", C.c_int), ("orc_get_part_score", C.c_float),
                          ("orc_get_num_connections", C.c_int), ("orc_get_num_candidates", C.c_int)]:
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
            getattr(L, name).restype = res
        L.orc_get_connection.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_Conn)]
        L.orc_get_candidate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_flip_average.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                       C.POINTER(C.c_float)]
        L.orc_find_peaks_plus.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.c_int]
        L.orc_find_peaks_3x3.argtypes = L.orc_find_peaks_plus.argtypes
        L.orc_heatmap_nms.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                      C.c_int, C.POINTER(C.c_int)]
        L.orc_resize_cubic.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_long, C.c_long, C.POINTER(C.c_float),
                                       C.c_int, C.c_int, C.c_long, C.c_long, C.c_double, C.c_double]
        L.orc_upsample4_planar_to_hwc.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.orc_pipeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
        L.orc_refine_centroid.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_double)]
        L.orc_py_find_humans.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        dp = C.POINTER(C.c_double)
        L.orc_predict_accumulate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, dp, dp]
        L.orc_find_peaks_original.argtypes = [dp, C.c_int, C.c_int, C.c_float, dp, C.c_int]
        L.orc_py_find_humans_f64.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.POINTER(C.c_int)]
        L.orc_resize_cubic_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_limb_pairs.restype = C.POINTER(C.c_int)
        L.orc_flip_heat_ord.restype = C.POINTER(C.c_int)
        L.orc_flip_paf_ord.restype = C.POINTER(C.c_int)
        self.L = L
        self.ctx = C.c_void_p(L.orc_create())

    def __del__(self):
        try:
            self.L.orc_destroy(self.ctx)
        except Exception:
            pass

    # ---- tables
    def tables(self):
        lp = np.array([self.L.orc_limb_pairs()[i] for i in range(60)]).reshape(30, 2)
        fh = np.array([self.L.orc_flip_heat_ord()[i] for i in range(20)])
        fp = np.array([self.L.orc_flip_paf_ord()[i] for i in range(30)])
        return lp, fh, fp

    # ---- stages
    def flip_average(self, net_out: np.ndarray, flip: bool = True):
        """net_out (2|1, 50, h, w) float16/float32 -> heat (20,h,w), paf (30,h,w) float32 planar."""
        net_out = np.ascontiguousarray(net_out)
        is_f16 = net_out.dtype == np.float16
        assert is_f16 or net_out.dtype == np.float32
        _, ch, h, w = net_out.shape
        assert ch == NUM_CH
        heat = np.empty((NUM_HEAT, h, w), np.float32)
        paf = np.empty((NUM_LIMB, h, w), np.float32)
        self.L.orc_flip_average(net_out.ctypes.data_as(C.c_void_p), int(is_f16), h, w, int(flip), _fp(heat), _fp(paf))
        return heat, paf

    def find_peaks(self, m: np.ndarray, thr: float = 0.1, mode: str = "plus"):
        m = np.ascontiguousarray(m, np.float32)
        h, w = m.shape
        xy = np.empty((h * w, 2), np.int32)
        fn = self.L.orc_find_peaks_plus if mode == "plus" else self.L.orc_find_peaks_3x3
        n = fn(_fp(m), h, w, thr, _ip(xy), h * w)
        return xy[:n].copy()

    def heatmap_nms(self, heat: np.ndarray, upsample: int = 4, refine: bool = True):
        """heat planar (>=18,h,w) -> joint_list (N,5) [x,y,score,id,part], part_count (18,)"""
        heat = np.ascontiguousarray(heat, np.float32)
        _, h, w = heat.shape
        cap = NUM_PART * h * w
        out = np.empty((cap, 5), np.float32)
        cnt = np.zeros(NUM_PART, np.int32)
        n = self.L.orc_heatmap_nms(_fp(heat), h, w, upsample, int(refine), _fp(out), cap, _ip(cnt))
        return out[:n].copy(), cnt

    def resize_cubic(self, src: np.ndarray, fx: float, fy: float):
        src = np.ascontiguousarray(src, np.float32)
        sh, sw = src.shape
        dh, dw = int(round(sh * fy)), int(round(sw * fx))
        dst = np.empty((dh, dw), np.float32)
        self.L.orc_resize_cubic(_fp(src), sh, sw, sw, 1, _fp(dst), dh, dw, dw, 1, 1.0 / fx, 1.0 / fy)
        return dst

    def upsample4_hwc(self, planar: np.ndarray):
        planar = np.ascontiguousarray(planar, np.float32)
        c, h, w = planar.shape
        dst = np.empty((4 * h, 4 * w, c), np.float32)
        self.L.orc_upsample4_planar_to_hwc(_fp(planar), c, h, w, _fp(dst))
        return dst

    def refine_centroid(self, m, x, y, radius=2):
        m = np.ascontiguousarray(m, np.float32)
        out = (C.c_double * 3)()
        self.L.orc_refine_centroid(_fp(m), m.shape[0], m.shape[1], x, y, radius, out)
        return tuple(out)

    def predict_accumulate(self, net_out, pad_down, pad_right, img_h, img_w, n_scales, heat_acc, paf_acc, flip=True):
        """one scale of predict(): accumulates into planar float64 heat_acc (20,H,W) / paf_acc (30,H,W) in place"""
        net_out = np.ascontiguousarray(net_out)
        is_f16 = net_out.dtype == np.float16
        _, _, h, w = net_out.shape
        dp = C.POINTER(C.c_double)
        self.L.orc_predict_accumulate(net_out.ctypes.data_as(C.c_void_p), int(is_f16), h, w, int(flip), pad_down, pad_right,
                                      img_h, img_w, n_scales, heat_acc.ctypes.data_as(dp), paf_acc.ctypes.data_as(dp))

    def find_peaks_original(self, heat_acc, thre1=0.1):
        _, H, W = heat_acc.shape
        cap = 18 * 4096
        rows = np.empty((cap, 5), np.float64)
        n = self.L.orc_find_peaks_original(heat_acc.ctypes.data_as(C.POINTER(C.c_double)), H, W, thre1,
                                           rows.ctypes.data_as(C.POINTER(C.c_double)), cap)
        return rows[:min(n, cap)].copy()

    def py_find_humans_f64(self, rows, paf_acc, img_height):
        """original path: rows (N,5) float64 from find_peaks_original, paf_acc planar (30,H,W) float64"""
        rows = np.ascontiguousarray(rows, np.float64).reshape(-1, 5)
        paf_acc = np.ascontiguousarray(paf_acc, np.float64)
        _, H, W = paf_acc.shape
        cap = 512
        out = np.empty((cap, 20, 2), np.float64)
        ncn = np.zeros(NUM_LIMB, np.int32)
        dp = C.POINTER(C.c_double)
        n = self.L.orc_py_find_humans_f64(rows.ctypes.data_as(dp), len(rows), paf_acc.ctypes.data_as(dp), H, W, int(img_height),
                                          out.ctypes.data_as(dp), cap, _ip(ncn))
        return out[:n].copy(), ncn

    def resize_u8(self, img, fx, fy):
        """cv2.resize(img, (0,0), fx=fx, fy=fy, INTER_CUBIC) for uint8 HWC (restated, unpinned)"""
        img = np.ascontiguousarray(img, np.uint8)
        sh, sw, cn = img.shape
        dh, dw = int(round(sh * fy)), int(round(sw * fx))
        out = np.empty((dh, dw, cn), np.uint8)
        self.L.orc_resize_cubic_u8(img.ctypes.data_as(C.c_void_p), sh, sw, cn, out.ctypes.data_as(C.c_void_p), dh, dw,
                                   1.0 / fx, 1.0 / fy)
        return out

    def py_find_humans(self, joint_list: np.ndarray, paf_hwc: np.ndarray, img_height: int):
        """find_connections + find_humans (the pure-Python twins): -> persons (P,20,2) float64, n_connections (30,)"""
        jl = np.ascontiguousarray(joint_list, np.float32).reshape(-1, 5)
        pm = np.ascontiguousarray(paf_hwc, np.float32)
        cap = 512
        out = np.empty((cap, 20, 2), np.float64)
        ncn = np.zeros(NUM_LIMB, np.int32)
        n = self.L.orc_py_find_humans(_fp(jl), len(jl), _fp(pm), pm.shape[0], pm.shape[1], pm.shape[2], int(img_height),
                                      out.ctypes.data_as(C.POINTER(C.c_double)), cap, _ip(ncn))
        return out[:n].copy(), ncn

    # ---- process_paf (same call shape as the reference's SWIG module)
    def process_paf(self, joint_list: np.ndarray, paf_hwc: np.ndarray, min_img_size: int):
        jl = np.ascontiguousarray(joint_list, np.float32)
        pm = np.ascontiguousarray(paf_hwc, np.float32)
        assert jl.ndim == 3 and pm.ndim == 3
        self.L.orc_process_paf(self.ctx, *jl.shape, _fp(jl), *pm.shape, _fp(pm), int(min_img_size))
        return self.result()

    def pipeline(self, net_out: np.ndarray, min_img_size: int, flip: bool = True):
        net_out = np.ascontiguousarray(net_out)
        is_f16 = net_out.dtype == np.float16
        _, ch, h, w = net_out.shape
        cap = 4096
        peaks = np.empty((cap, 5), np.float32)
        n = C.c_int(0)
        self.L.orc_pipeline(self.ctx, net_out.ctypes.data_as(C.c_void_p), int(is_f16), h, w, int(flip),
                            int(min_img_size), _fp(peaks), cap, C.byref(n))
        res = self.result() if n.value > 0 else {"ids": np.zeros((0, 18), np.int32), "scores": np.zeros(0, np.float32),
                                                 "peaks": np.zeros((0, 3), np.float32), "connections": [[] for _ in range(30)],
                                                 "n_candidates": np.zeros(30, np.int32),
                                                 "candidates": [[] for _ in range(30)], "sort_oob": False}
        res["joint_list"] = peaks[:min(n.value, cap)].copy()
        return res

    def result(self):
        L, ctx = self.L, self.ctx
        nh = L.orc_get_num_humans(ctx)
        ids = np.array([[L.orc_get_part_peak_id(ctx, s, p) for p in range(NUM_PART)] for s in range(nh)],
                       np.int32).reshape(nh, NUM_PART)
        scores = np.array([L.orc_get_score(ctx, s) for s in range(nh)], np.float32)
        npk = L.orc_get_num_peaks(ctx)
        peaks = np.array([[L.orc_get_part_x(ctx, i), L.orc_get_part_y(ctx, i), L.orc_get_part_score(ctx, i)]
                          for i in range(npk)], np.float32).reshape(npk, 3)
        conns = []
        for limb in range(NUM_LIMB):
            lst = []
            for i in range(L.orc_get_num_connections(ctx, limb)):
                cn = _Conn()
                L.orc_get_connection(ctx, limb, i, C.byref(cn))
                lst.append((cn.cid1, cn.cid2, cn.score, cn.peak_id1, cn.peak_id2, cn.length))
            conns.append(lst)
        ncand = np.array([L.orc_get_num_candidates(ctx, l) for l in range(NUM_LIMB)], np.int32)
        cands = []
        i1, i2, sc, ov, ln = C.c_int(), C.c_int(), C.c_float(), C.c_float(), C.c_float()
        for limb in range(NUM_LIMB):
            lst = []
            for i in range(int(ncand[limb])):
                L.orc_get_candidate(ctx, limb, i, C.byref(i1), C.byref(i2), C.byref(sc), C.byref(ov), C.byref(ln))
                lst.append((i1.value, i2.value, sc.value, ov.value, ln.value))
            cands.append(lst)
        return {"ids": ids, "scores": scores, "peaks": peaks, "connections": conns, "n_candidates": ncand,
                "candidates": cands, "sort_oob": bool(L.orc_get_sort_oob(ctx))}


class Reference:
    """The reference's compiled C++ (file-scope globals: one instance per process is meaningful)."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            build()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (needs /root/reference to build; it travels prebuilt to the GPU box)")
        L = C.CDLL(REF_SO)
        self.process_paf_ = getattr(L, "_Z11process_pafiiiPfiiiS_i")
        self.process_paf_.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_float), C.c_int]
        self.get_num_humans = getattr(L, "_Z14get_num_humansv")
        self.get_part_peak_id = getattr(L, "_Z16get_part_peak_idii")
        self.get_part_peak_id.argtypes = [C.c_int, C.c_int]
        self.get_score = getattr(L, "_Z9get_scorei")
        self.get_score.argtypes = [C.c_int]
        self.get_score.restype = C.c_float
        self.get_part_x = getattr(L, "_Z10get_part_xi")
        self.get_part_y = getattr(L, "_Z10get_part_yi")
        self.get_part_score = getattr(L, "_Z14get_part_scorei")
        self.get_part_score.restype = C.c_float
        for f in (self.get_part_x, self.get_part_y, self.get_part_score):
            f.argtypes = [C.c_int]

    def process_paf(self, joint_list: np.ndarray, paf_hwc: np.ndarray, min_img_size: int):
        jl = np.ascontiguousarray(joint_list, np.float32)
        pm = np.ascontiguousarray(paf_hwc, np.float32)
        self.process_paf_(*jl.shape, _fp(jl), *pm.shape, _fp(pm), int(min_img_size))
        nh = self.get_num_humans()
        ids = np.array([[self.get_part_peak_id(s, p) for p in range(NUM_PART)] for s in range(nh)],
                       np.int32).reshape(nh, NUM_PART)
        scores = np.array([self.get_score(s) for s in range(nh)], np.float32)
        npk = jl.shape[0] * jl.shape[1]
        peaks = np.array([[self.get_part_x(i), self.get_part_y(i), self.get_part_score(i)] for i in range(npk)],
                         np.float32).reshape(npk, 3)
        return {"ids": ids, "scores": scores, "peaks": peaks}
