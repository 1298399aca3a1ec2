"""Host-side wrapper of the native batched path (pp_process_batch / pp_nms_batch).

torch is used for device memory and streams only; every number comes out of the HIP kernels."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PP_F16, PP_F32, RECORD_BYTES, RECORD_DTYPE, PosePafError


class PosePostProcessor:
    """Flip-average -> NMS/refine -> limb scoring -> matching -> assembly for a batch of network outputs that
    already live in HBM.  Mirrors evaluate.py:75-129 (--run_refactor --run_cpp) per image."""

    def __init__(self, max_batch: int = 64, max_h: int = 128, max_w: int = 128, max_peaks_per_part: int = 64,
                 device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise PosePafError("no HIP device: the post-processing path has no CPU implementation")
        self.L = _lib.load()
        self.ctx = C.c_void_p()
        torch.cuda.set_device(device)
        _lib.check(self.L.pp_create(C.byref(self.ctx), device, max_batch, max_h, max_w, max_peaks_per_part))
        self.device = device
        self.max_batch, self.maxp = max_batch, max_peaks_per_part
        self._records = torch.empty(max_batch * RECORD_BYTES, dtype=torch.uint8, device=f"cuda:{device}")

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.L.pp_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _dtype_code(t):
        import torch
        if t.dtype == torch.float16:
            return PP_F16
        if t.dtype == torch.float32:
            return PP_F32
        raise PosePafError(f"network output must be float16 or float32, got {t.dtype}")

    def _check_input(self, net_out, flip):
        if net_out.dim() != 5 or net_out.shape[2] != _lib.NUM_CH or net_out.shape[1] != (2 if flip else 1):
            raise PosePafError(f"expected (B, {2 if flip else 1}, 50, h, w), got {tuple(net_out.shape)}")
        if not net_out.is_cuda or not net_out.is_contiguous():
            raise PosePafError("network output must be a contiguous device tensor")
        return net_out.shape[0], net_out.shape[3], net_out.shape[4]

    def process_async(self, net_out, min_img_size: int = 512, flip: bool = True, min_img_size_dev=None, records=None):
        """Enqueue K_A, K_B, K_C on torch's current stream.  Returns the device record buffer
        (uint8 tensor, B * sizeof(pp_record)); no host synchronisation."""
        import torch
        B, h, w = self._check_input(net_out, flip)
        rec = self._records if records is None else records
        stream = torch.cuda.current_stream(net_out.device).cuda_stream
        mis = C.c_void_p(min_img_size_dev.data_ptr()) if min_img_size_dev is not None else None
        _lib.check(self.L.pp_process_batch(self.ctx, B, C.c_void_p(net_out.data_ptr()), self._dtype_code(net_out), h, w,
                                           int(flip), int(min_img_size), mis, C.c_void_p(rec.data_ptr()),
                                           C.c_void_p(stream)), self.ctx)
        return rec[: B * RECORD_BYTES]

    def process_py_async(self, net_out, img_height: int = 512, flip: bool = True, img_height_dev=None, records=None):
        """The pure-Python twins' rules (find_connections + find_humans) instead of the C++ pafprocess rules; enqueues
        on torch's current stream and returns the device record buffer."""
        import torch
        B, h, w = self._check_input(net_out, flip)
        rec = self._records if records is None else records
        stream = torch.cuda.current_stream(net_out.device).cuda_stream
        _lib.check(self.L.pp_process_batch_py(self.ctx, B, C.c_void_p(net_out.data_ptr()), self._dtype_code(net_out), h, w,
                                              int(flip), int(img_height),
                                              C.c_void_p(img_height_dev.data_ptr()) if img_height_dev is not None else None,
                                              C.c_void_p(rec.data_ptr()),
                                              C.c_void_p(stream)), self.ctx)
        return rec[: B * RECORD_BYTES]

    def process_py(self, net_out, img_height: int = 512, flip: bool = True) -> np.ndarray:
        return records_to_numpy(self.process_py_async(net_out, img_height, flip))

    def process(self, net_out, min_img_size: int = 512, flip: bool = True) -> np.ndarray:
        """Blocking form: returns a numpy structured array of B records (RECORD_DTYPE)."""
        rec = self.process_async(net_out, min_img_size, flip)
        return records_to_numpy(rec)

    def nms(self, net_out, flip: bool = True, refine: bool = True):
        """Config-2 path: joint lists only.  Returns list over images of (N,5) [x,y,score,id,part]."""
        import torch
        B, h, w = self._check_input(net_out, flip)
        stream = torch.cuda.current_stream(net_out.device).cuda_stream
        _lib.check(self.L.pp_nms_batch(self.ctx, B, C.c_void_p(net_out.data_ptr()), self._dtype_code(net_out), h, w,
                                       int(flip), int(refine), None, None, C.c_void_p(stream)), self.ctx)
        return [self.read_peaks(i) for i in range(B)]

    def nms_ex(self, net_out, flip: bool = True, nms_mode: int = 0, threshold: float = 0.1, refine_mode: int = 1):
        """pp_nms_batch_ex: the original path's 3x3 / >= NMS (nms_mode 1) and refine_centroid (refine_mode 2)."""
        import torch
        B, h, w = self._check_input(net_out, flip)
        stream = torch.cuda.current_stream(net_out.device).cuda_stream
        _lib.check(self.L.pp_nms_batch_ex(self.ctx, B, C.c_void_p(net_out.data_ptr()), self._dtype_code(net_out), h, w,
                                          int(flip), int(nms_mode), float(threshold), int(refine_mode), None, None,
                                          C.c_void_p(stream)), self.ctx)
        return [self.read_peaks(i) for i in range(B)]

    def time_kernels(self, net_out, min_img_size: int = 512, flip: bool = True, iters: int = 20):
        """HIP-event timing of each kernel on torch's current stream: dict name -> ms per launch."""
        import torch
        B, h, w = self._check_input(net_out, flip)
        ms = (C.c_float * 4)()
        stream = torch.cuda.current_stream(net_out.device).cuda_stream
        _lib.check(self.L.pp_time_kernels(self.ctx, B, C.c_void_p(net_out.data_ptr()), self._dtype_code(net_out), h, w,
                                          int(flip), int(min_img_size), int(iters), ms, C.c_void_p(stream)), self.ctx)
        return {"k_heat_peaks": ms[0], "k_limb_connect": ms[1], "k_assemble_wave": ms[2], "chain": ms[3]}

    def set_mode(self, mode: int):
        """pp_debug_set_mode: 0 fused + load-ordered (default), 1 separate assembly launch, 2 fused without ordering"""
        _lib.check(self.L.pp_debug_set_mode(self.ctx, int(mode)), self.ctx)

    def read_peaks(self, image: int) -> np.ndarray:
        cap = _lib.NUM_PART * self.maxp
        buf = np.empty((cap, 5), np.float32)
        n = C.c_int(0)
        _lib.check(self.L.pp_read_peaks(self.ctx, image, buf.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)), self.ctx)
        return buf[: n.value].copy()

    def read_part_counts(self, image: int) -> np.ndarray:
        buf = (C.c_int * _lib.NUM_PART)()
        _lib.check(self.L.pp_read_part_counts(self.ctx, image, buf), self.ctx)
        return np.array(buf[:], np.int32)

    def read_connection_counts(self, image: int) -> np.ndarray:
        buf = (C.c_int * _lib.NUM_LIMB)()
        _lib.check(self.L.pp_read_connection_counts(self.ctx, image, buf), self.ctx)
        return np.array(buf[:], np.int32)

    def debug_read_flags(self, image: int) -> np.ndarray:
        buf = (C.c_uint32 * (_lib.NUM_PART + _lib.NUM_LIMB))()
        _lib.check(self.L.pp_debug_read_flags(self.ctx, image, buf), self.ctx)
        return np.array(buf[:], np.uint32)

    def read_connections(self, image: int, limb: int) -> np.ndarray:
        buf = np.empty((self.maxp, 4), np.float32)
        n = C.c_int(0)
        _lib.check(self.L.pp_read_connections(self.ctx, image, limb, buf.ctypes.data_as(C.POINTER(C.c_float)), self.maxp,
                                              C.byref(n)), self.ctx)
        return buf[: n.value].copy()


def check_status(recs, allow=0, what="records") -> int:
    """OR of the status words of a batch of records; raises PosePafError when a bit outside include/posepaf.h:53-59
    is set (corruption) or when a bit outside `allow` is set."""
    st = int(np.bitwise_or.reduce(np.asarray(recs["status"], dtype=np.uint32))) if len(recs) else 0
    if st & ~_lib.ST_DEFINED_MASK:
        raise PosePafError(f"{what}: undefined bits in pp_record.status: {st:#010x}")
    if st & ~allow:
        raise PosePafError(f"{what}: status flags {st & ~allow:#x} raised (allowed: {allow:#x})")
    return st


def records_to_numpy(rec_u8) -> np.ndarray:
    """device/host uint8 tensor of packed pp_record -> numpy structured array (synchronises)."""
    host = rec_u8.cpu().numpy()
    return host.view(RECORD_DTYPE).copy()


def record_humans(rec) -> list:
    """One record -> list of dicts {ids (18,), x (18,), y (18,), part_score (18,), score, n_parts}."""
    out = []
    for i in range(int(rec["n_humans"])):
        h = rec["humans"][i]
        out.append({"ids": h["peak_id"].copy(), "x": h["x"].copy(), "y": h["y"].copy(),
                    "part_score": h["part_score"].copy(), "score": float(h["score"]), "n_parts": int(h["n_parts"])})
    return out
