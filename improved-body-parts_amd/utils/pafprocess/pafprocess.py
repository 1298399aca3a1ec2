"""Same module-level functions as the reference's SWIG-generated `pafprocess.py`
(utils/pafprocess/pafprocess.i:14 + pafprocess.h:70-76), bound to libposepaf.so with ctypes.

    pafprocess.process_paf(joint_list, paf_upsamp, img_h)      # evaluate.py:108-110
    pafprocess.get_num_humans(); get_part_peak_id(h, p); get_score(h)
    pafprocess.get_part_x(cid); get_part_y(cid); get_part_score(cid)

Argument handling mirrors the `IN_ARRAY3` typemap (numpy.i:1097-1126): each array argument must be
3-dimensional and is converted to a contiguous float32 array (copied if needed); anything else raises
TypeError/ValueError like the SWIG wrapper.  The computation runs on the GPU; a missing library or device raises
(there is no CPU path)."""
import ctypes as _C

import numpy as _np

from posepaf import _lib as _l


def _as_array3(a, name):
    try:
        arr = _np.ascontiguousarray(a, dtype=_np.float32)
    except Exception as e:  # SWIG: "array of type float32 required"
        raise TypeError(f"{name}: cannot convert to a float32 array ({e})")
    if arr.ndim != 3:
        raise TypeError(f"Array must have 3 dimensions.  Given array has {arr.ndim} dimensions")
    return arr


def process_paf(peaks, pafmap, min_img_size):
    L = _l.load()
    pk = _as_array3(peaks, "peaks")
    pm = _as_array3(pafmap, "pafmap")
    fp = _C.POINTER(_C.c_float)
    rc = L.process_paf(pk.shape[0], pk.shape[1], pk.shape[2], pk.ctypes.data_as(fp), pm.shape[0], pm.shape[1], pm.shape[2],
                       pm.ctypes.data_as(fp), int(min_img_size))
    _l.check(rc)
    return rc


def get_num_humans():
    return _l.load().get_num_humans()


def get_part_peak_id(skeleton_id, part_id):
    return _l.load().get_part_peak_id(int(skeleton_id), int(part_id))


def get_score(skeleton_id):
    return _l.load().get_score(int(skeleton_id))


def get_part_x(cid):
    return _l.load().get_part_x(int(cid))


def get_part_y(cid):
    return _l.load().get_part_y(int(cid))


def get_part_score(cid):
    return _l.load().get_part_score(int(cid))
