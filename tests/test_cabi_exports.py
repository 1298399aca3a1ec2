"""CPU: libposepaf.so builds for gfx950, loads, and exports every symbol include/posepaf.h declares.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    from posepaf import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.LIB_PATH


def test_header_symbols_exported(lib_path):
    hdr = open(os.path.join(ROOT, "include", "posepaf.h")).read()
    declared = re.findall(r"^PP_API\s+[\w\s\*]+?\b(\w+)\(", hdr, flags=re.M)
    assert len(declared) >= 25
    for ref_name in ("process_paf", "get_num_humans", "get_part_peak_id", "get_score", "get_part_x", "get_part_y",
                     "get_part_score"):
        assert ref_name in declared  # utils/pafprocess/pafprocess.h:70-76
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/posepaf.h but not exported"
    from posepaf import _lib
    assert sorted(_lib.EXPORTS) == sorted(declared)


def test_record_layout_matches_header():
    """numpy view of pp_record must match the C struct (checked by compiling a sizeof probe)."""
    import subprocess
    import tempfile
    from posepaf import _lib
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "posepaf.h"\nint main(){printf("%zu %zu %zu %zu\\n",' \
          'sizeof(pp_human),sizeof(pp_record),offsetof(pp_record,humans),offsetof(pp_human,score));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")],
                       check=True)
        out = subprocess.run([os.path.join(d, "p")], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == _lib.HUMAN_DTYPE.itemsize
    assert int(out[1]) == _lib.RECORD_DTYPE.itemsize
    assert int(out[2]) == _lib.RECORD_DTYPE.fields["humans"][1]
    assert int(out[3]) == _lib.HUMAN_DTYPE.fields["score"][1]


def test_no_device_is_loud(lib_path):
    """Without a GPU every compute entry point must refuse, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from posepaf import _lib
    L = _lib.load()
    assert L.pp_device_available() == 0
    ctx = ctypes.c_void_p()
    assert L.pp_create(ctypes.byref(ctx), 0, 1, 128, 128, 64) == -1  # PP_ERR_NO_DEVICE
    import numpy as np
    from utils.pafprocess import pafprocess
    with pytest.raises(_lib.PosePafError):
        pafprocess.process_paf(np.zeros((1, 1, 5), np.float32), np.zeros((8, 8, 30), np.float32), 8)
    with pytest.raises(TypeError):
        pafprocess.process_paf(np.zeros((1, 5), np.float32), np.zeros((8, 8, 30), np.float32), 8)
    from posepaf.api import PosePostProcessor
    with pytest.raises(_lib.PosePafError):
        PosePostProcessor()


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "improved-body-parts_amd")
    offenders = []
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                for m in re.finditer(r"^\s*(from|import)\s+oracle\b|posepaf_oracle|liboracle|oracle/_ref|oracle\.oracle", txt, re.M):
                    line = txt[: m.start()].count("\n") + 1
                    context = txt.splitlines()[line - 1].strip()
                    if context.startswith(("#", "//", "*", "/*")) or "see oracle/" in context:
                        continue
                    offenders.append(f"{fn}:{line}: {context}")
    assert not offenders, offenders
