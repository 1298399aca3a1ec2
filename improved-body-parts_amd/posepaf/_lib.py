"""ctypes binding of libposepaf.so (include/posepaf.h).  The library IS the product path: if it is missing
or does not load, importing this module's users fails loudly -- there is no Python/CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POSEPAF_LIB") or os.path.join(_HERE, "libposepaf.so")  # override: A/B builds of the library

NUM_PART, NUM_LIMB, NUM_HEAT, NUM_CH = 18, 30, 20, 50
MAX_HUMANS = 128
PP_F32, PP_F16 = 0, 1

ST_PEAK_OVERFLOW, ST_HUMAN_OVERFLOW, ST_SKEL_OVERFLOW, ST_SORT_UNDEFINED, ST_CAND_OVERFLOW, ST_FLOAT_COORDS = 1, 2, 4, 8, 16, 32
ST_SYNC_TIMEOUT = 64
ST_DEFINED_MASK = 0x7F            # include/posepaf.h:53-61; any other bit in pp_record.status is corruption
ST_OVERFLOW_MASK = ST_PEAK_OVERFLOW | ST_HUMAN_OVERFLOW | ST_SKEL_OVERFLOW | ST_CAND_OVERFLOW

# numpy views of pp_human / pp_record (include/posepaf.h)
HUMAN_DTYPE = np.dtype([("peak_id", "<i4", (NUM_PART,)), ("x", "<i4", (NUM_PART,)), ("y", "<i4", (NUM_PART,)),
                        ("part_score", "<f4", (NUM_PART,)), ("score", "<f4"), ("n_parts", "<i4")])
RECORD_DTYPE = np.dtype([("n_humans", "<i4"), ("n_peaks", "<i4"), ("status", "<u4"), ("n_connections", "<i4"),
                         ("humans", HUMAN_DTYPE, (MAX_HUMANS,))])
RECORD_BYTES = RECORD_DTYPE.itemsize

EXPORTS = [
    "pp_create", "pp_destroy", "pp_last_hip_error", "pp_status_string", "pp_device_available", "pp_process_batch", "pp_process_batch_py",
    "pp_nms_batch", "pp_nms_batch_ex", "pp_time_kernels", "pp_bias_act_f16", "pp_maxpool2_f16", "pp_upsample2_f16", "pp_add3_f16", "pp_channel_mean_f16", "pp_channel_mean_finish_f16", "pp_nhwc64_to_planes_f16", "pp_se_gains_f16", "pp_conv_own_sums_splits", "pp_conv_own_sums_f16", "pp_channel_scale_f16", "pp_preprocess_u8", "pp_preprocess_u8_ragged", "pp_flip_average", "pp_pwconv_supported", "pp_pwconv_f16", "pp_conv_num_configs", "pp_conv_f16", "pp_conv_ld_f16", "pp_conv_own_supported", "pp_conv_own_f16", "pp_conv_own_ld_f16", "pp_conv_own_ex_f16", "pp_stem7x7_f16", "pp_conv_up2_collapsed_f16", "pp_pw_supported", "pp_pw_f16", "pp_pw_pool_f16", "pp_pw_cat_f16", "pp_conv_debug_clock", "pp_debug_set_stamps", "pp_debug_set_mode", "pp_read_peaks", "pp_read_connections", "pp_read_part_counts", "pp_read_connection_counts", "pp_debug_read_flags", "pp_read_records", "pp_process_paf_host",
    "pp_get_num_humans", "pp_get_part_peak_id", "pp_get_score", "pp_get_part_x", "pp_get_part_y",
    "pp_get_part_score", "pp_get_status", "pp_py_find_connections_host", "pp_py_find_humans_host", "pp_original_accumulate", "pp_original_accumulate_all", "pp_original_finish",
    "pp_resize_u8_cubic",
    # the reference's seven names (utils/pafprocess/pafprocess.h:70-76)
    "process_paf", "get_num_humans", "get_part_peak_id", "get_score", "get_part_x", "get_part_y", "get_part_score",
]


class PosePafError(RuntimeError):
    pass


_lib = None


def build(verbose: bool = False) -> str:
    """Compile libposepaf.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    csrc = os.path.join(os.path.dirname(_HERE), "csrc")
    r = subprocess.run(["make", "-C", csrc], capture_output=True, text=True)
    if r.returncode != 0:
        raise PosePafError("building libposepaf.so failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PosePafError(f"{LIB_PATH} is missing: build it with `make -C improved-body-parts_amd/csrc` "
                           "(or __graft_entry__.build()).  There is no CPU fallback.")
    try:  # share torch's HIP runtime (same SONAME) when torch is in the process, so device pointers are common
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.pp_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.pp_destroy.argtypes = [vp]
    L.pp_last_hip_error.argtypes = [vp]
    L.pp_status_string.argtypes = [C.c_int]
    L.pp_status_string.restype = C.c_char_p
    L.pp_process_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.pp_process_batch_py.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.pp_nms_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.pp_nms_batch_ex.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp]
    L.pp_time_kernels.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, vp]
    L.pp_bias_act_f16.argtypes = [vp, vp, vp, vp, C.c_long, C.c_int, C.c_float, C.c_int, vp]
    L.pp_maxpool2_f16.argtypes = [vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, vp]
    L.pp_upsample2_f16.argtypes = [vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, vp]
    L.pp_add3_f16.argtypes = [vp, vp, vp, vp, C.c_long, vp]
    L.pp_channel_mean_f16.argtypes = [vp, vp, vp, C.c_int, C.c_long, C.c_int, C.c_int, vp]
    L.pp_channel_mean_finish_f16.argtypes = [vp, vp, C.c_int, C.c_long, C.c_int, C.c_int, vp]
    L.pp_conv_own_sums_splits.argtypes = [C.c_int, C.c_int]
    L.pp_conv_own_sums_f16.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    L.pp_channel_scale_f16.argtypes = [vp, vp, vp, C.c_int, C.c_long, C.c_int, vp]
    L.pp_preprocess_u8.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.pp_preprocess_u8_ragged.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.pp_flip_average.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.pp_pwconv_supported.argtypes = [C.c_int, C.c_int]
    L.pp_pwconv_f16.argtypes = [vp, vp, vp, vp, vp, vp, C.c_long, C.c_int, C.c_int, C.c_float, C.c_int, vp]
    L.pp_conv_num_configs.argtypes = []
    L.pp_conv_f16.argtypes = [vp, vp, vp, vp, vp] + [C.c_int] * 9 + [C.c_float, C.c_int, vp]
    L.pp_conv_ld_f16.argtypes = [vp, vp, vp, vp, vp] + [C.c_int] * 9 + [C.c_float, C.c_int, C.c_int, C.c_int, vp]
    L.pp_conv_own_f16.argtypes = [vp, vp, vp, vp, vp] + [C.c_int] * 9 + [C.c_float, C.c_int, vp]
    L.pp_se_gains_f16.argtypes = [vp] * 7 + [C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    L.pp_nhwc64_to_planes_f16.argtypes = [vp, vp, C.c_int, C.c_long, C.c_int, vp]
    L.pp_conv_own_ld_f16.argtypes = [vp, vp, vp, vp, vp] + [C.c_int] * 9 + [C.c_float, C.c_int, C.c_int, C.c_int, vp]
    L.pp_conv_own_ex_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 9 + [C.c_float, C.c_int, C.c_int, vp]
    L.pp_pw_supported.argtypes = [C.c_int, C.c_int]
    L.pp_pw_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    L.pp_pw_pool_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, vp]
    L.pp_conv_up2_collapsed_f16.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.c_int, vp]
    L.pp_pw_cat_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_float, vp]
    L.pp_stem7x7_f16.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    L.pp_conv_own_supported.argtypes = [C.c_int, C.c_int, C.c_int]
    L.pp_debug_set_stamps.argtypes = [vp]
    L.pp_debug_set_mode.argtypes = [vp, C.c_int]
    L.pp_read_peaks.argtypes = [vp, C.c_int, fp, C.c_int, ip]
    L.pp_read_connections.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int, ip]
    L.pp_read_records.argtypes = [vp, vp, vp, C.c_int, vp]
    L.pp_read_connection_counts.argtypes = [vp, C.c_int, ip]
    L.pp_read_part_counts.argtypes = [vp, C.c_int, ip]
    L.pp_debug_read_flags.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32)]
    L.pp_process_paf_host.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int]
    dp = C.POINTER(C.c_double)
    L.pp_py_find_connections_host.argtypes = [vp, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, ip]
    L.pp_py_find_humans_host.argtypes = [vp, dp, ip, fp, C.c_int, dp, C.c_int, ip]
    L.pp_original_accumulate.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, vp, vp, vp, vp, vp]
    L.pp_original_accumulate_all.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.c_int, ip, ip, C.c_int, ip, ip, C.c_int, C.c_int,
                                             vp, vp, vp]
    L.pp_original_finish.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp, vp]
    L.pp_resize_u8_cubic.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]
    L.pp_get_num_humans.argtypes = [vp]
    L.pp_get_part_peak_id.argtypes = [vp, C.c_int, C.c_int]
    L.pp_get_score.argtypes = [vp, C.c_int]
    L.pp_get_score.restype = C.c_float
    L.pp_get_part_x.argtypes = [vp, C.c_int]
    L.pp_get_part_y.argtypes = [vp, C.c_int]
    L.pp_get_part_score.argtypes = [vp, C.c_int]
    L.pp_get_part_score.restype = C.c_float
    L.pp_get_status.argtypes = [vp]
    L.pp_get_status.restype = C.c_uint32
    L.process_paf.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int]
    L.get_part_peak_id.argtypes = [C.c_int, C.c_int]
    L.get_score.argtypes = [C.c_int]
    L.get_score.restype = C.c_float
    L.get_part_x.argtypes = [C.c_int]
    L.get_part_y.argtypes = [C.c_int]
    L.get_part_score.argtypes = [C.c_int]
    L.get_part_score.restype = C.c_float
    _lib = L
    return L


def check(rc: int, ctx=None):
    if rc != 0:
        L = load()
        msg = L.pp_status_string(rc).decode()
        if ctx is not None and rc == -4:
            msg += f" (hipError {L.pp_last_hip_error(ctx)})"
        raise PosePafError(f"libposepaf: {msg} [{rc}]")
