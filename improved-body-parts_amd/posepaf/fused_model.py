"""Inference-time form of the IMHN for MI355X: BatchNorm folded into the convolutions, fp16, channels-last
(NHWC is also the layout the images arrive in, utils/parse_skeletons.py:60-73, so the input permute is free),
dead heads removed, whole forward replayed from a HIP graph.

Every stride-1 convolution with channel counts that are multiples of 8 runs as ONE fused kernel of libposepaf.so -- the
hand-written LDS-DMA + MFMA kernels of csrc/posepaf_conv_own.hip (3x3 halo-tile kernel incl. the hourglass' upsample and adds,
implicit-GEMM kernel) or a composable_kernel template with this library's epilogue (csrc/posepaf_conv_inst.hip), whichever the
per-shape timing picks; the choice table can be saved, loaded, hashed and broadcast (save_table / load_table / table_hash).
Only the 7x7 stride-2 stem and the 50-channel heads stay on PyTorch-ROCm (MIOpen / hipBLASLt).  Structural changes:
  * BN (eval) folded:  w' = w * g / sqrt(var + eps),  b' = beta - mean * g / sqrt(var + eps)
  * only `[-1][0]` (last stage, full-resolution scale) is produced -- the sole output the inference path reads
    (utils/parse_skeletons.py:80); the last stage's coarse-scale feature/pred heads feed nothing and are skipped
  * LeakyReLU applied in place on the conv output

`FusedIMHN.from_network(NetworkEval)` takes the weights of the checkpoint-compatible definition in
models/posenet.py, so a reference checkpoint loads there (strict=True) and is then folded here.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

LEAK = 0.01
# Fused MFMA kernel for the supported 1x1 convolutions (csrc/posepaf_pwconv.hip).  Measured on MI355X (tools/pwconv_probe.py,
# 64x128x128 batch): 2.1-3.2 TB/s, i.e. 0.83-1.21x of MIOpen's CK kernel + the separate epilogue pass, which both already
# stream near the HBM rate -- not a win yet at one 4-wave workgroup per CU, so it stays off by default.
USE_PWCONV = False
# Fused convolution + epilogue in one kernel (csrc/posepaf_conv_inst.hip: composable_kernel's XDL implicit-GEMM main loop with
# this project's bias / LeakyReLU / residual / post-add functor on the fp32 accumulators).  Per layer shape the tile
# configurations AND the MIOpen-convolution + separate-epilogue path are timed once, at first (eager) use, and the fastest is kept.
USE_FUSED_CONV = True
# Hand-written implicit-GEMM / halo-tile convolution kernels (csrc/posepaf_conv_own.hip) compete in the same per-shape timing:
# configuration ids >= 100 -> workgroup tile: 256 pixels x 256 / 128 / 64 channels (implicit GEMM), or 512 = the 3x3 halo-tile kernel
USE_OWN_CONV = True
TUNE_MIOPEN = os.environ.get("POSEPAF_TUNE_MIOPEN", "0") == "1"   # also time MIOpen + epilogue pass where fused kernels exist
OWN_VARIANTS = {101: 256, 102: 128, 103: 64, 104: 512}
PW_VARIANT = 105            # the streaming 1x1 kernel (pp_pw_f16): weights resident in LDS, pixel fragments straight from HBM
USE_PW = True
USE_SUM_FUSION = True       # the SE block's channel sums leave the 3x3 kernel that produces the feature map
USE_SLICE_OUTPUT = True     # the backbone's concatenation is written in place by its two producers
USE_COLLAPSED_UP2 = True    # conv3x3(upsample2(x)) as four 2x2 convolutions of x (2.25x fewer multiply-adds)
USE_FOLDED_MERGE = True     # merge_preds(head(f)) is linear in f: folded into merge_features' weights at load time
USE_CAT_SKIP = True         # a residual block's last 1x1 and its 1x1 skip convolution as one product over [t ; x]
USE_SE_KERNEL = True        # the SE block's excitation (two tiny linear layers, LeakyReLU, sigmoid) in one launch per block
USE_POOL_FUSION = True      # the hourglass' 2x2 max-pools leave the 1x1 kernel that produces their input as a second output
_conv_choice: dict = {}   # shape key -> tile configuration id, or -1 = MIOpen convolution + k_bias_act pass
_conv_timing: dict = {}   # shape key -> {"miopen": ms, cfg: ms, ...} measured by the autotune (diagnostics)
_conv_calls: dict = {}    # shape key -> number of forward() calls since import (diagnostics)
_TUNE_REPS = 5
OWN_MARGIN = 0.03           # timing noise band inside which the hand-written kernel is preferred over a library template


_progress = None   # callable(str) or None: one line per tuned layer shape (bench.py prints them to stderr: a silent warm-up of
                   # several minutes looks like a hang to a job supervisor)


def set_progress(fn) -> None:
    global _progress
    _progress = fn


def _timed(fn):
    """median of _TUNE_REPS single-call timings after one warm-up (HIP events on the current stream)"""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(_TUNE_REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def _note(key, choice) -> None:
    if _progress is not None:
        _progress(f"tuned shape {len(_conv_choice)}: {key} -> {choice}")


def conv_choices() -> dict:
    """The per-shape decisions taken so far (for bench.py / DESIGN.md)."""
    return dict(_conv_choice)


# ---- the kernel-choice table is an artefact, not a side effect: it can be saved, loaded, hashed and broadcast, so that two
# runs (or the ranks of one job) execute the SAME kernel per layer shape and produce the same fp16 bits.
def _key_to_json(k):
    return [("b", bool(v)) if isinstance(v, bool) else v for v in k]


def _key_from_json(k):
    return tuple(bool(v[1]) if isinstance(v, list) else v for v in k)


def table_entries() -> list:
    """sorted [[key, choice], ...] in a JSON-able form"""
    return sorted(([_key_to_json(k), int(v)] for k, v in _conv_choice.items()), key=lambda e: repr(e[0]))


def table_hash() -> str:
    import hashlib
    import json
    return hashlib.sha256(json.dumps(table_entries()).encode()).hexdigest()[:16]


def library_hash() -> str:
    """identifies the kernels a table was tuned on: sha256 of libposepaf.so"""
    import hashlib
    from . import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def default_table_path() -> str:
    d = os.environ.get("POSEPAF_CACHE_DIR", "/tmp/posepaf_cache")
    return os.path.join(d, f"conv_choice_{library_hash()}.json")


def save_table(path: str | None = None) -> str:
    import json
    path = path or default_table_path()
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump({"library": library_hash(), "entries": table_entries()}, f)
    os.replace(tmp, path)
    return path


def install_entries(entries) -> int:
    """merge [[key, choice], ...] into the table (existing keys keep their choice); -> number of new keys"""
    n = 0
    for k, v in entries:
        k = _key_from_json(k)
        if k not in _conv_choice:
            _conv_choice[k] = int(v)
            n += 1
    return n


def load_table(path: str | None = None, force: bool = False) -> int:
    """-> number of entries installed (0 when the file is absent or was tuned on another build of the library)"""
    import json
    path = path or default_table_path()
    if not os.path.exists(path):
        return 0
    try:
        doc = json.load(open(path))
    except Exception:
        return 0
    if doc.get("library") != library_hash() and not force:
        return 0
    return install_entries(doc.get("entries", []))


def _cl(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def to_planes(t: torch.Tensor) -> torch.Tensor:
    """The contiguous (n, c, h, w) copy of a prediction: the [:, :c] view of the last head's 64-channel pixel-major output goes
    through pp_nhwc64_to_planes_f16 (one streaming pass); anything else through torch's strided copy."""
    n, c, h, w = t.shape
    if (t.is_cuda and t.dtype == torch.float16 and c <= 64 and t.stride() == (h * w * 64, 1, w * 64, 64)
            and t.data_ptr() % 16 == 0 and n <= 65535):
        from . import _lib
        y = torch.empty((n, c, h, w), dtype=t.dtype, device=t.device)
        _lib.check(_lib.load().pp_nhwc64_to_planes_f16(_ptr(t), _ptr(y), n, h * w, c, _stream(t)))
        return y
    return t.contiguous()


def _slice_ld(t):
    """pixel stride (elements) of t when it is channels-last OR a channel slice of a channels-last tensor, else None"""
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if sc == 1 and sw >= c and sh == w * sw and (n == 1 or sn == h * w * sw) and sw % 8 == 0:
        return sw
    return None


def _ptr(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t):
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_fallback_seen = set()


def _fallback(name: str, x: torch.Tensor, why: str) -> None:
    """A DEVICE tensor that a HIP helper does not take runs on the torch operator instead: say so once per (helper, reason)
    (POSEPAF_STRICT=1: raise).  Host tensors (CPU unit tests of the module structure) are the torch path by design."""
    if not x.is_cuda:
        return
    msg = f"posepaf.fused_model.{name}: torch fallback on a device tensor ({why}; dtype {x.dtype}, shape {tuple(x.shape)})"
    if os.environ.get("POSEPAF_STRICT", "0") == "1":
        raise RuntimeError(msg)
    key = (name, why)
    if key not in _fallback_seen:
        _fallback_seen.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def hip_bias_act_(y: torch.Tensor, bias: torch.Tensor, res, act: bool, post=None) -> torch.Tensor:
    """In place y = act(y + bias[c] (+ res)) (+ post) by the fused HIP epilogue kernel (csrc/posepaf_epilogue.hip)."""
    from . import _lib
    y = _cl(y)
    res = _cl(res) if res is not None else None
    post = _cl(post) if post is not None else None
    rc = _lib.load().pp_bias_act_f16(_ptr(y), _ptr(bias), _ptr(res), _ptr(post), y.numel(), y.shape[1], LEAK, int(act),
                                     _stream(y))
    _lib.check(rc)
    return y


def maxpool2(x: torch.Tensor) -> torch.Tensor:
    """2x2/2 max pool; HIP kernel on channels-last fp16, torch elsewhere."""
    n, c, h, w = x.shape
    if not (x.is_cuda and x.dtype == torch.float16 and c % 8 == 0 and h % 2 == 0 and w % 2 == 0):
        _fallback("maxpool2", x, "needs fp16, channels % 8 == 0, even height and width")
        return F.max_pool2d(x, 2, 2)
    from . import _lib
    x = _cl(x)
    y = torch.empty((n, c, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.load().pp_maxpool2_f16(_ptr(x), _ptr(y), n, h // 2, w // 2, c, _stream(x)))
    return y


def upsample2(x: torch.Tensor) -> torch.Tensor:
    """nearest x2 upsample; HIP kernel on channels-last fp16, torch elsewhere."""
    n, c, h, w = x.shape
    if not (x.is_cuda and x.dtype == torch.float16 and c % 8 == 0):
        _fallback("upsample2", x, "needs fp16 and channels % 8 == 0")
        return F.interpolate(x, scale_factor=2, mode="nearest")
    from . import _lib
    x = _cl(x)
    y = torch.empty((n, c, h * 2, w * 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.load().pp_upsample2_f16(_ptr(x), _ptr(y), n, h, w, c, _stream(x)))
    return y


def add3(a: torch.Tensor, b: torch.Tensor, c=None) -> torch.Tensor:
    """a + b (+ c) in one pass; HIP kernel on channels-last fp16, torch elsewhere."""
    if not (a.is_cuda and a.dtype == torch.float16 and a.numel() % 8 == 0 and a.shape == b.shape
            and (c is None or c.shape == a.shape)):
        _fallback("add3", a, "needs fp16 tensors of one shape with a multiple of 8 elements")
        return a + b if c is None else a + b + c
    from . import _lib
    a, b = _cl(a), _cl(b)
    c = _cl(c) if c is not None else None
    y = torch.empty_like(a, memory_format=torch.channels_last)
    _lib.check(_lib.load().pp_add3_f16(_ptr(a), _ptr(b), _ptr(c), _ptr(y), a.numel(), _stream(a)))
    return y


def channel_mean(x: torch.Tensor) -> torch.Tensor:
    """(n, c, h, w) -> (n, c) mean over the pixels (the SE squeeze); HIP kernel on channels-last fp16, torch elsewhere."""
    n, c, h, w = x.shape
    if not (x.is_cuda and x.dtype == torch.float16 and c % 8 == 0 and c <= 2048):
        _fallback("channel_mean", x, "needs fp16 and channels % 8 == 0, <= 2048")
        return x.mean(dim=(2, 3))
    from . import _lib
    x = _cl(x)
    splits = max(1, min(h * w, -(-1024 // n)))
    ws = torch.empty((n, splits, c), dtype=torch.float32, device=x.device)
    out = torch.empty((n, c), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().pp_channel_mean_f16(_ptr(x), _ptr(ws), _ptr(out), n, h * w, c, splits, _stream(x)))
    return out


def channel_scale(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """x * s[:, :, None, None] (the SE excitation); HIP kernel on channels-last fp16, torch elsewhere."""
    n, c, h, w = x.shape
    if not (x.is_cuda and x.dtype == torch.float16 and s.dtype == torch.float16 and c % 8 == 0):
        _fallback("channel_scale", x, "needs fp16 and channels % 8 == 0")
        return x * s[:, :, None, None]
    from . import _lib
    x = _cl(x)
    s = s.contiguous()
    y = torch.empty_like(x, memory_format=torch.channels_last)
    _lib.check(_lib.load().pp_channel_scale_f16(_ptr(x), _ptr(s), _ptr(y), n, h * w, c, _stream(x)))
    return y


def _use_hip(x: torch.Tensor, c_out: int) -> bool:
    return x.is_cuda and x.dtype == torch.float16 and c_out % 8 == 0


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d | None):
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None:
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * s.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
    return w, b


class Scaled:
    """An activation together with its SE gains (models/layers_transposed.py:289-310: out = y * s[n, c]) NOT yet multiplied:
    the 1x1 convolutions that consume it multiply in their input read (pp_pw_f16's `scale`), so the separate x * s pass over
    the feature map is never made.  materialize() is that pass, for consumers that cannot fold it."""

    def __init__(self, y, s):
        self.y, self.s = y, s

    def materialize(self):
        return channel_scale(self.y, self.s)


class FConv(nn.Module):
    """conv + folded-BN bias [+ LeakyReLU]"""

    def __init__(self, conv: nn.Conv2d, bn, act: bool):
        super().__init__()
        w, b = _fold(conv, bn)
        self.weight = nn.Parameter(w, requires_grad=False)
        self.bias = nn.Parameter(b, requires_grad=False)
        self.stride, self.padding, self.dilation = conv.stride, conv.padding, conv.dilation
        self.act = act

    def conv_only(self, x):
        return F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation)

    def _pointwise_ok(self, x):
        if not (USE_PWCONV and x.is_cuda and x.dtype == torch.float16 and self.weight.shape[2:] == (1, 1)
                and self.stride == (1, 1) and self.padding == (0, 0)):
            return False
        from . import _lib
        return bool(_lib.load().pp_pwconv_supported(self.weight.shape[1], self.weight.shape[0]))

    # ---- fused convolution (pp_conv_f16)
    def _fused_eligible(self, x, res, post):
        return (USE_FUSED_CONV and x.is_cuda and x.dtype == torch.float16 and self.stride == (1, 1)
                and self.weight.shape[2] == self.weight.shape[3] and self.padding[0] == self.padding[1]
                and self.dilation[0] == self.dilation[1] and self.weight.shape[0] % 8 == 0 and self.weight.shape[1] % 8 == 0
                and not (res is not None and post is not None))

    def _fused_launch(self, cfg, x, extra, mode, y):
        from . import _lib
        n, c, h, w = x.shape
        sliced = x.stride(3) != c or y.stride(3) != self.weight.shape[0]
        if cfg == 104 and sliced:   # the 3x3 halo kernel takes pixel strides on both sides
            return _lib.load().pp_conv_own_ld_f16(_ptr(x), _ptr(self.weight), _ptr(self.bias), _ptr(extra), _ptr(y), n, h, w, c,
                                                  self.weight.shape[0], self.weight.shape[2], self.padding[0], self.dilation[0], mode,
                                                  LEAK if self.act else 1.0, 512, x.stride(3), y.stride(3), _stream(x))
        if cfg >= 100 and (x.stride(3) != c or (cfg != PW_VARIANT and y.stride(3) != self.weight.shape[0])):
            return -6   # the implicit-GEMM kernels read and write packed pixels; the streaming 1x1 kernel writes a channel slice
        if cfg == PW_VARIANT:
            if self.weight.shape[2] != 1 or self.padding[0] != 0:
                return -6
            return _lib.load().pp_pw_f16(_ptr(x), None, _ptr(self.weight), _ptr(self.bias), _ptr(extra), None, _ptr(y), None,
                                         n * h * w, h * w, c, self.weight.shape[0], y.stride(3), mode, LEAK if self.act else 1.0,
                                         _stream(x))
        if cfg >= 100:   # hand-written kernels (csrc/posepaf_conv_own.hip): workgroup-tile variant OWN_VARIANTS[cfg]
            return _lib.load().pp_conv_own_f16(_ptr(x), _ptr(self.weight), _ptr(self.bias), _ptr(extra), _ptr(y), n, h, w, c,
                                               self.weight.shape[0], self.weight.shape[2], self.padding[0], self.dilation[0], mode,
                                               LEAK if self.act else 1.0, OWN_VARIANTS[cfg], _stream(x))
        return _lib.load().pp_conv_ld_f16(_ptr(x), _ptr(self.weight), _ptr(self.bias), _ptr(extra), _ptr(y), n, h, w, c,
                                          self.weight.shape[0], self.weight.shape[2], self.padding[0], self.dilation[0], mode,
                                          LEAK if self.act else 1.0, cfg, x.stride(3), y.stride(3), _stream(x))

    def _fused(self, x, res, post, out=None):
        """-> y, or None when this shape runs faster (or only) on the MIOpen + epilogue path.  x may be a channel slice of a wider
        channels-last tensor and `out` (optional) another: the kernels that take pixel strides read / write them in place."""
        from . import _lib
        x = x if _slice_ld(x) is not None else _cl(x)
        extra = res if res is not None else post
        extra = _cl(extra) if extra is not None else None
        mode = 1 if res is not None else (2 if post is not None else 0)
        if not self.weight.is_contiguous(memory_format=torch.channels_last):
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        n, c, h, w = x.shape
        k, r = self.weight.shape[0], self.weight.shape[2]
        ho = h + 2 * self.padding[0] - self.dilation[0] * (r - 1)
        wo = w + 2 * self.padding[0] - self.dilation[0] * (r - 1)
        key = (n, c, h, w, k, r, self.padding[0], self.dilation[0], mode, bool(self.act))
        sliced = x.stride(3) != c or (out is not None and out.stride(3) != k)
        if sliced:   # its own tuning entry: only the kernels that take pixel strides compete
            key = key + ("slice", x.stride(3), out.stride(3) if out is not None else k)
        choice = _conv_choice.get(key)
        _conv_calls[key] = _conv_calls.get(key, 0) + 1
        y = out if out is not None else torch.empty((n, k, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                return None  # cannot time inside a capture; shapes are tuned by the eager warm-up pass
            choice = self._tune(key, x, extra, mode, y, res, post)
        if choice < 0:
            return None
        _lib.check(self._fused_launch(choice, x, extra, mode, y))
        return y

    def _tune(self, key, x, extra, mode, y, res, post):
        from . import _lib
        L = _lib.load()

        def timed(fn):   # median of _TUNE_REPS single-launch timings after one warm-up
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(_TUNE_REPS):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2]

        best, best_t, times = -1, float("inf"), {}
        own = [c_ for c_ in OWN_VARIANTS if USE_OWN_CONV and L.pp_conv_own_supported(x.shape[1], self.weight.shape[0],
                                                                                     self.weight.shape[2])]
        if USE_PW and USE_OWN_CONV and self.weight.shape[2] == 1 and L.pp_pw_supported(x.shape[1], self.weight.shape[0]):
            own = own + [PW_VARIANT]
        for cfg in list(range(L.pp_conv_num_configs())) + own:
            if self._fused_launch(cfg, x, extra, mode, y) != 0:
                continue
            t = timed(lambda: self._fused_launch(cfg, x, extra, mode, y))
            times[cfg] = t
            # (the library templates are timed first; a hand-written kernel within OWN_MARGIN of the best of them takes the shape --
            # the 1x1 shapes sit within +-2 % of each other and would otherwise flip between runs)
            if t < best_t * (1.0 + OWN_MARGIN if cfg >= 100 and 0 <= best < 100 else 1.0):
                best, best_t = cfg, t
        # The MIOpen convolution + separate epilogue pass is the fallback of shapes no fused kernel takes.  Timing it where fused
        # kernels exist costs an exhaustive MIOpen find per shape (most of the warm-up) and it never won a shape worth more than
        # 0.06 ms: it is timed only on request (POSEPAF_TUNE_MIOPEN=1) or when nothing else ran.
        if best < 0 or TUNE_MIOPEN:
            t = timed(lambda: hip_bias_act_(self.conv_only(x), self.bias, res, self.act, post))
            times["miopen"] = t
            if t < best_t:
                best, best_t = -1, t
        _conv_choice[key] = best
        _conv_timing[key] = times
        _note(key, best)
        return best

    # ---- x2 nearest upsample in front, up to two tensors added behind: one launch of the 3x3 halo kernel (pp_conv_own_ex_f16)
    def _collapsed_weights(self):
        """The 3x3 weights behind a x2 nearest upsample, collapsed per output phase (py, px) into 2x2 weights on the
        half-resolution grid: an input pixel is reached through the SUM of the taps that land on it (rows: py = 0 ->
        {w[0], w[1] + w[2]}, py = 1 -> {w[0] + w[1], w[2]}; columns alike).  (4, K, 2, 2, C) fp16, sums in fp32, rounded once."""
        w4 = getattr(self, "_w4", None)
        if w4 is None or w4.device != self.weight.device:
            w = self.weight.detach().float()                                   # (K, C, 3, 3)
            rows = [[w[:, :, 0], w[:, :, 1] + w[:, :, 2]], [w[:, :, 0] + w[:, :, 1], w[:, :, 2]]]   # [py][a] -> (K, C, 3)
            out = []
            for py in range(2):
                for px in range(2):
                    taps = []
                    for a in range(2):
                        r = rows[py][a]                                         # (K, C, 3): columns still separate
                        cols = [r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if px == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]]
                        taps.append(torch.stack(cols, dim=1))                  # (K, 2 [b], C)
                    out.append(torch.stack(taps, dim=1))                       # (K, 2 [a], 2 [b], C)
            w4 = torch.stack(out).to(self.weight.dtype).contiguous()          # (4, K, 2, 2, C)
            self._w4 = w4
        return w4

    def forward_up2(self, low, post, post2=None):
        """act(conv(upsample2(low)) + bias) + post (+ post2), three ways, the fastest kept per shape (timed once):
        0 separate: upsample2 -> convolution (-> add3);  1 the upsample read through the 3x3 halo kernel's own loads, adds in its
        epilogue (pp_conv_own_ex_f16);  2.. the COLLAPSED form (pp_conv_up2_collapsed_f16): four 2x2 convolutions of the
        half-resolution tensor, 2.25x fewer multiply-adds for the same real-number result (choice = 2 + the tile width index)."""
        from . import _lib
        n, c, h, w = low.shape
        k = self.weight.shape[0]
        key = ("up2", n, c, h, w, k, post2 is not None, bool(self.act))
        fused_ok = (USE_OWN_CONV and low.is_cuda and low.dtype == torch.float16 and self.stride == (1, 1)
                    and tuple(self.weight.shape[2:]) == (3, 3) and self.padding == (1, 1) and self.dilation == (1, 1))
        COLLAPSED_BN = (256, 128, 64, 512)   # 512: the halo-tile kernel walking the four taps of each output phase

        def separate():
            if post2 is None:
                return self(upsample2(low), post=post)
            return add3(self(upsample2(low)), post, post2)   # the conv runs without the extra read, ONE pass adds the three

        def fused():
            x = _cl(low)
            if not self.weight.is_contiguous(memory_format=torch.channels_last):
                self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
            y = torch.empty((n, k, 2 * h, 2 * w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            e1, e2 = _cl(post), (_cl(post2) if post2 is not None else None)
            rc = _lib.load().pp_conv_own_ex_f16(_ptr(x), _ptr(self.weight), _ptr(self.bias), _ptr(e1), _ptr(e2), _ptr(y), None, n, 2 * h, 2 * w,
                                                c, k, 3, 1, 1, 3 if post2 is not None else 2, LEAK if self.act else 1.0, 512, 1, _stream(x))
            return y if rc == 0 else None

        def collapsed(bn):
            if not USE_COLLAPSED_UP2 or k % (bn if bn != 512 else 64):
                return None
            x = _cl(low)
            y = torch.empty((n, k, 2 * h, 2 * w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            e1, e2 = _cl(post), (_cl(post2) if post2 is not None else None)
            rc = _lib.load().pp_conv_up2_collapsed_f16(_ptr(x), _ptr(self._collapsed_weights()), _ptr(self.bias), _ptr(e1), _ptr(e2), _ptr(y),
                                                       n, h, w, c, k, 3 if post2 is not None else 2, LEAK if self.act else 1.0, bn,
                                                       _stream(x))
            return y if rc == 0 else None

        choice = _conv_choice.get(key) if fused_ok else 0
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                return separate()
            separate()                      # tunes the inner convolution's shape first
            times = {"separate": _timed(separate)}
            choice, best = 0, times["separate"]
            if fused() is not None:
                times["fused"] = _timed(fused)
                if times["fused"] < best:
                    choice, best = 1, times["fused"]
            for i, bn in enumerate(COLLAPSED_BN):
                if collapsed(bn) is not None:
                    times[f"collapsed{bn}"] = _timed(lambda: collapsed(bn))
                    if times[f"collapsed{bn}"] < best:
                        choice, best = 2 + i, times[f"collapsed{bn}"]
            _conv_timing[key] = times
            _conv_choice[key] = choice
            _note(key, choice)
        if choice >= 2:
            y = collapsed(COLLAPSED_BN[choice - 2])
            if y is not None:
                return y
        elif choice == 1:
            y = fused()
            if y is not None:
                return y
        return separate()

    # ---- two outputs: y = act(conv(x) + bias + res) and y + other, one launch of an own kernel (pp_conv_own_ex_f16 mode 4)
    def forward_dual(self, x, res, other, want_pool: bool = False):
        """-> (y, y + other) with y = act(conv(x) + bias + res).  One launch with two stores when that beats the fused
        convolution followed by a tensor add, timed once per shape.  x may be a Scaled pair (activation, SE gains)."""
        from . import _lib
        scale = None
        if isinstance(x, Scaled):
            x, scale = x.y, x.s
        n, c, h, w = x.shape
        k, r = self.weight.shape[0], self.weight.shape[2]
        pool_ok = want_pool and USE_POOL_FUSION and h % 2 == 0 and w % (64 if c == 64 else 32) == 0
        key = ("dual", n, c, h, w, k, r, self.padding[0], self.dilation[0], bool(self.act), scale is not None, pool_ok)
        if res is None:
            key = key + ("nores",)
        ok = (USE_OWN_CONV and x.is_cuda and x.dtype == torch.float16 and self.stride == (1, 1)
              and self.weight.shape[2] == self.weight.shape[3] and self.padding[0] == self.padding[1] and self.dilation[0] == self.dilation[1]
              and 2 * self.padding[0] == self.dilation[0] * (r - 1) and _lib.load().pp_conv_own_supported(c, k, r))

        def separate():
            y = self(x if scale is None else channel_scale(x, scale), res)
            y2 = y + other
            return (y, y2, maxpool2(y2)) if want_pool else (y, y2)

        def fused(bn):
            if res is None and bn != PW_VARIANT:
                return None                  # only the streaming 1x1 kernel has the two-output form without a residual (mode 5)
            xx, rr, oo = _cl(x), (_cl(res) if res is not None else None), _cl(other)
            dmode = 4 if res is not None else 5
            if not self.weight.is_contiguous(memory_format=torch.channels_last):
                self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
            y = torch.empty((n, k, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            y2 = torch.empty_like(y, memory_format=torch.channels_last)
            if bn == PW_VARIANT:   # the streaming 1x1 kernel, optionally with the SE gains of `x` folded into its input read
                if not (USE_PW and r == 1 and self.padding[0] == 0 and _lib.load().pp_pw_supported(c, k)):
                    return None
                if pool_ok:   # ... and the 2x2 max-pool of y2 (the next stage's hourglass pools its input first)
                    pooled = torch.empty((n, k, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
                    rc = _lib.load().pp_pw_pool_f16(_ptr(xx), _ptr(scale), _ptr(self.weight), _ptr(self.bias), _ptr(rr), _ptr(oo),
                                                    _ptr(y), _ptr(y2), _ptr(pooled), n * h * w, h * w, w, c, k, k, dmode,
                                                    LEAK if self.act else 1.0, _stream(x))
                    return (y, y2, pooled) if rc == 0 else None
                rc = _lib.load().pp_pw_f16(_ptr(xx), _ptr(scale), _ptr(self.weight), _ptr(self.bias), _ptr(rr), _ptr(oo), _ptr(y),
                                           _ptr(y2), n * h * w, h * w, c, k, k, dmode, LEAK if self.act else 1.0, _stream(x))
                if rc != 0:
                    return None
                return (y, y2, maxpool2(y2)) if want_pool else (y, y2)
            if scale is not None:
                return None
            rc = _lib.load().pp_conv_own_ex_f16(_ptr(xx), _ptr(self.weight), _ptr(self.bias), _ptr(rr), _ptr(oo), _ptr(y), _ptr(y2), n, h, w,
                                                c, k, r, self.padding[0], self.dilation[0], 4, LEAK if self.act else 1.0, bn, 0, _stream(x))
            if rc != 0:
                return None
            return (y, y2, maxpool2(y2)) if want_pool else (y, y2)

        choice = _conv_choice.get(key) if ok else 0
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                return separate()
            separate()                      # tunes the plain convolution's shape first

            def timed(fn):
                fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(_TUNE_REPS):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return sorted(ts)[len(ts) // 2]
            times = {"separate": timed(separate)}
            choice, best = 0, times["separate"]
            for bn in (256, 128, 64, 512, PW_VARIANT):
                if bn != PW_VARIANT and k % (bn if bn != 512 else 128):
                    continue
                if fused(bn) is None:
                    continue
                times[bn] = timed(lambda: fused(bn))
                if times[bn] < best:
                    choice, best = bn, times[bn]
            _conv_timing[key] = times
            _conv_choice[key] = choice
            _note(key, choice)
        if choice:
            out = fused(choice)
            if out is not None:
                return out
        return separate()

    def forward_scaled(self, xs: "Scaled", res=None):
        """act(conv1x1(xs.y * xs.s) + bias (+ res)) with the gains folded into the kernel's input read; None when not taken"""
        from . import _lib
        x, sc = _cl(xs.y), xs.s.contiguous()
        n, c, h, w = x.shape
        k = self.weight.shape[0]
        if not (USE_PW and USE_OWN_CONV and x.is_cuda and x.dtype == torch.float16 and tuple(self.weight.shape[2:]) == (1, 1)
                and self.stride == (1, 1) and self.padding == (0, 0) and (h * w) % 64 == 0 and _lib.load().pp_pw_supported(c, k)):
            return None
        if not self.weight.is_contiguous(memory_format=torch.channels_last):
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        y = torch.empty((n, k, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        res = _cl(res) if res is not None else None
        rc = _lib.load().pp_pw_f16(_ptr(x), _ptr(sc), _ptr(self.weight), _ptr(self.bias), _ptr(res), None, _ptr(y), None, n * h * w,
                                   h * w, c, k, k, 1 if res is not None else 0, LEAK if self.act else 1.0, _stream(x))
        return y if rc == 0 else None

    def forward_mean(self, x, partial: bool = False):
        """-> (y, channel mean of y (n, c_out)) with y = act(conv(x) + bias): the SE squeeze of models/layers_transposed.py:298-303.
        On the 3x3 halo-tile kernel the per-tile channel sums leave the convolution's epilogue (pp_conv_own_sums_f16) and only a
        tiny reduction follows; timed once per shape against convolution + the two-pass channel mean.
        partial=True: the fused form hands over the partial sums themselves, (ws (n, splits, c_out) fp32, splits, h * w), for a
        consumer that finishes them on the way (FSE / pp_se_gains_f16)."""
        from . import _lib
        n, c, h, w = x.shape
        k = self.weight.shape[0]
        key = ("mean", n, c, h, w, k, bool(self.act))
        L = _lib.load() if x.is_cuda else None
        splits = L.pp_conv_own_sums_splits(h, w) if L is not None else 0
        ok = (USE_OWN_CONV and USE_SUM_FUSION and x.is_cuda and x.dtype == torch.float16 and self.stride == (1, 1) and splits > 0
              and tuple(self.weight.shape[2:]) == (3, 3) and self.padding == (1, 1) and self.dilation == (1, 1)
              and c % 32 == 0 and k % 64 == 0)

        def separate():
            y = self(x)
            return y, channel_mean(y)

        def fused(hand_over=False):
            xx = _cl(x)
            if not self.weight.is_contiguous(memory_format=torch.channels_last):
                self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
            y = torch.empty((n, k, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            ws = torch.empty((n, splits, k), dtype=torch.float32, device=x.device)
            rc = L.pp_conv_own_sums_f16(_ptr(xx), _ptr(self.weight), _ptr(self.bias), _ptr(y), _ptr(ws), n, h, w, c, k,
                                        LEAK if self.act else 1.0, _stream(x))
            if rc != 0:
                return None
            if hand_over:
                return y, (ws, splits, h * w)
            mean = torch.empty((n, k), dtype=x.dtype, device=x.device)
            _lib.check(L.pp_channel_mean_finish_f16(_ptr(ws), _ptr(mean), n, h * w, k, splits, _stream(x)))
            return y, mean

        choice = _conv_choice.get(key) if ok else 0
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                return separate()
            separate()
            if fused() is None:
                choice = 0
            else:
                t_sep, t_fused = _timed(separate), _timed(fused)
                _conv_timing[key] = {"separate": t_sep, "fused": t_fused}
                choice = 1 if t_fused < t_sep else 0
            _conv_choice[key] = choice
            _note(key, choice)
        if choice:
            out = fused(partial)
            if out is not None:
                return out
        return separate()

    def forward_pool(self, x, res=None):
        """-> (y, maxpool2(y)) with y = act(conv(x) + bias (+ res)).  For a 1x1 convolution the pooled tensor can leave the
        streaming kernel as one more output (pp_pw_pool_f16) instead of a pass of its own; timed once per shape against
        convolution + k_maxpool2."""
        from . import _lib
        n, c, h, w = x.shape
        k = self.weight.shape[0]
        key = ("pool", n, c, h, w, k, res is not None, bool(self.act))
        ok = (USE_PW and USE_POOL_FUSION and USE_OWN_CONV and x.is_cuda and x.dtype == torch.float16 and self.stride == (1, 1)
              and tuple(self.weight.shape[2:]) == (1, 1) and self.padding == (0, 0) and h % 2 == 0
              and w % (64 if c == 64 else 32) == 0 and _lib.load().pp_pw_supported(c, k))

        def separate():
            y = self(x, res)
            return y, maxpool2(y)

        def fused():
            xx = _cl(x)
            rr = _cl(res) if res is not None else None
            if not self.weight.is_contiguous(memory_format=torch.channels_last):
                self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
            y = torch.empty((n, k, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            pooled = torch.empty((n, k, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            rc = _lib.load().pp_pw_pool_f16(_ptr(xx), None, _ptr(self.weight), _ptr(self.bias), _ptr(rr), None, _ptr(y), None,
                                            _ptr(pooled), n * h * w, h * w, w, c, k, k, 1 if res is not None else 0,
                                            LEAK if self.act else 1.0, _stream(x))
            return (y, pooled) if rc == 0 else None

        choice = _conv_choice.get(key) if ok else 0
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                return separate()
            separate()                      # tunes the plain convolution's shape first
            if fused() is None:
                choice = 0
            else:
                t_sep, t_fused = _timed(separate), _timed(fused)
                _conv_timing[key] = {"separate": t_sep, "fused": t_fused}
                choice = 1 if t_fused < t_sep else 0
            _conv_choice[key] = choice
            _note(key, choice)
        if choice:
            out = fused()
            if out is not None:
                return out
        return separate()

    def forward(self, x, res=None, post=None, out=None):
        """act(conv(x) + bias (+ res)) (+ post); out: optional destination, e.g. a channel slice of a wider channels-last tensor"""
        if out is not None:
            if not isinstance(x, Scaled) and self._fused_eligible(x, res, post) and _slice_ld(out) is not None:
                y = self._fused(x, res, post, out)
                if y is not None:
                    return y
            out.copy_(self.forward(x if isinstance(x, Scaled) or _slice_ld(x) == x.shape[1] else x.contiguous(memory_format=torch.channels_last), res, post))
            return out
        if isinstance(x, Scaled):
            y = self.forward_scaled(x, res) if post is None else None
            if y is not None:
                return y
            x = x.materialize()
        if self._fused_eligible(x, res, post):
            y = self._fused(x, res, post)
            if y is not None:
                return y
        if self._pointwise_ok(x):   # 1x1: one MFMA GEMM with the epilogue fused (csrc/posepaf_pwconv.hip)
            from . import _lib
            x = _cl(x)
            n, k, h, w = x.shape
            cout = self.weight.shape[0]
            y = torch.empty((n, cout, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            res = _cl(res) if res is not None else None
            post = _cl(post) if post is not None else None
            _lib.check(_lib.load().pp_pwconv_f16(_ptr(x), _ptr(self.weight), _ptr(self.bias), _ptr(res), _ptr(post), _ptr(y),
                                                 n * h * w, k, cout, LEAK, int(self.act), _stream(x)))
            return y
        if _use_hip(x, self.weight.shape[0]):
            return hip_bias_act_(self.conv_only(x), self.bias, res, self.act, post)
        y = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)
        if res is not None:
            y = y + res
        if self.act:
            y = F.leaky_relu_(y, LEAK)
        return y + post if post is not None else y


class FStem(FConv):
    """The 7x7 / stride 2 stem (models/layers_transposed.py:78-87) on the one-pass HIP kernel pp_stem7x7_f16; the weights are
    re-laid once into the kernel's k order (7 rows x 8-pixel window x 3 channels, zero column on the left, padded to 192)."""

    def __init__(self, conv, bn, act):
        super().__init__(conv, bn, act)
        w = self.weight.detach()                                   # (64, 3, 7, 7), BN folded
        wp = torch.zeros((w.shape[0], 7, 8, 3), dtype=w.dtype)
        wp[:, :, 1:, :] = w.permute(0, 2, 3, 1)                    # [k][r][s + 1][c]
        prepared = torch.zeros((w.shape[0], 192), dtype=w.dtype)
        prepared[:, :168] = wp.reshape(w.shape[0], 168)
        self.prepared = nn.Parameter(prepared, requires_grad=False)

    def forward(self, x, res=None, post=None):
        n, c, h, w = x.shape
        ok = (x.is_cuda and x.dtype == torch.float16 and res is None and post is None and c == 3 and self.weight.shape[0] == 64
              and tuple(self.weight.shape[2:]) == (7, 7) and self.stride == (2, 2) and self.padding == (3, 3) and h % 2 == 0
              and w % 4 == 0 and x.permute(0, 2, 3, 1).is_contiguous())
        if not ok:
            _fallback("stem", x, "needs an fp16 NHWC image with even height and width % 4 == 0")
            return super().forward(x, res, post)
        from . import _lib
        y = torch.empty((n, 64, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _lib.check(_lib.load().pp_stem7x7_f16(_ptr(x), _ptr(self.prepared), _ptr(self.bias), _ptr(y), n, h, w,
                                              LEAK if self.act else 1.0, _stream(x)))
        return y


class FHead(FConv):
    """The 1x1 prediction heads (256 -> 50 channels, models/posenet.py:60-66).  50 is not a multiple of the kernels' 64-channel
    granule: weights and bias are zero-padded to 64 outputs once, the streaming 1x1 kernel writes a 64-channel tensor whose
    channels 50..63 are exact zeros, and the caller gets the [:, :50] view of it."""

    def __init__(self, conv, bn, act):
        super().__init__(conv, bn, act)
        k, c = self.weight.shape[:2]
        kp = -(-k // 64) * 64
        wpad = torch.zeros((kp, c, 1, 1), dtype=self.weight.dtype)
        wpad[:k] = self.weight.detach()
        bpad = torch.zeros(kp, dtype=self.bias.dtype)
        bpad[:k] = self.bias.detach()
        self.wpad = nn.Parameter(wpad, requires_grad=False)
        self.bpad = nn.Parameter(bpad, requires_grad=False)
        self.k_real = k
        self.last_padded = None

    def forward(self, x, res=None, post=None):
        from . import _lib
        xs = x if isinstance(x, Scaled) else None
        xx = _cl(xs.y if xs is not None else x)
        n, c, h, w = xx.shape
        kp = self.wpad.shape[0]
        ok = (USE_PW and USE_OWN_CONV and res is None and post is None and xx.is_cuda and xx.dtype == torch.float16
              and tuple(self.weight.shape[2:]) == (1, 1) and (xs is None or (h * w) % 64 == 0) and _lib.load().pp_pw_supported(c, kp))
        if ok:
            if not self.wpad.is_contiguous(memory_format=torch.channels_last):
                self.wpad.data = self.wpad.data.contiguous(memory_format=torch.channels_last)
            y = torch.empty((n, kp, h, w), dtype=xx.dtype, device=xx.device, memory_format=torch.channels_last)
            rc = _lib.load().pp_pw_f16(_ptr(xx), _ptr(xs.s.contiguous()) if xs is not None else None, _ptr(self.wpad), _ptr(self.bpad),
                                       None, None, _ptr(y), None, n * h * w, h * w, c, kp, kp, 0, LEAK if self.act else 1.0,
                                       _stream(xx))
            if rc == 0:
                self.last_padded = y           # the 64-channel tensor behind the view (channels 50..63 are exact zeros)
                return y[:, : self.k_real]
        self.last_padded = None
        return super().forward(xs.materialize() if xs is not None else x, res, post)


def _fconv_from_block(m, act=None):  # models.layers_transposed.Conv / DilatedConv
    return FConv(m.conv, m.bn, m.relu is not None if act is None else act)


class FResidual(nn.Module):
    def __init__(self, r):
        super().__init__()
        cb = r.convBlock
        self.c1 = FConv(cb[0], cb[1], True)
        self.c2 = FConv(cb[3], cb[4], True)
        self.c3 = FConv(cb[6], cb[7], False)
        self.c3.act = r.relu_flag  # the block's final LeakyReLU is applied by c3's epilogue, after the skip add
        self.skip = FConv(r.skipConv[0], r.skipConv[1], False) if r.ins != r.outs else None
        if self.skip is not None:  # conv3 + b3 + convs + bs: one bias vector, the skip conv runs bias-free
            self.c3.bias = nn.Parameter(self.c3.bias + self.skip.bias, requires_grad=False)

    def forward(self, x, out=None):
        if self.skip is not None and out is None:
            y = self._cat(x, False)
            if y is not None:
                return y[0]
        res = self.skip.conv_only(x) if self.skip is not None else x
        return self.c3(self.c2(self.c1(x)), res, out=out)

    def forward_pool(self, x):
        """-> (block output, its 2x2 max-pool): the pooled tensor leaves the block's last 1x1 convolution as a second output"""
        if self.skip is not None:
            y = self._cat(x, True)
            if y is not None:
                return y
        res = self.skip.conv_only(x) if self.skip is not None else x
        return self.c3.forward_pool(self.c2(self.c1(x)), res)

    def _cat(self, x, want_pool):
        """conv3(t) + skip(x) as ONE product over the concatenated channels [t ; x] (pp_pw_cat_f16) when both weight matrices fit
        the streaming kernel's LDS in one piece (otherwise the input would be read once per output-channel split and nothing is
        gained); timed once per shape against the unfused form.  -> (y, pooled or None), or None."""
        from . import _lib
        if not (USE_PW and USE_CAT_SKIP and USE_OWN_CONV and x.is_cuda and x.dtype == torch.float16):
            return None
        k, c1 = self.c3.weight.shape[:2]
        c2 = self.skip.weight.shape[1]
        n, _, h, w = x.shape
        # (weights beyond 144 KB split the output channels over workgroup columns that read the same pixels at about the same
        # time -- the re-reads are served by the cache hierarchy; the timing below decides)
        if (tuple(self.skip.weight.shape[2:]) != (1, 1) or self.skip.stride != (1, 1)
                or not _lib.load().pp_pw_supported(c1 + c2, k) or c1 % 32 or c2 % 32):
            return None
        # (beyond 512 input channels the kernel works on 16-pixel groups, which have no 2-row form to pool)
        pool_ok = want_pool and USE_POOL_FUSION and h % 2 == 0 and w % (64 if c1 + c2 == 64 else 32) == 0 and c1 + c2 <= 512
        key = ("cat", n, c1, c2, h, w, k, bool(self.c3.act), pool_ok)
        if getattr(self, "_wcat", None) is None or self._wcat.device != x.device:
            self._wcat = torch.cat([self.c3.weight.detach().flatten(1), self.skip.weight.detach().flatten(1)], dim=1).contiguous()

        def separate():
            res = self.skip.conv_only(x)
            t = self.c2(self.c1(x))
            return self.c3.forward_pool(t, res) if want_pool else (self.c3(t, res), None)

        def fused():
            t = _cl(self.c2(self.c1(x)))
            xx = _cl(x)
            y = torch.empty((n, k, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            pooled = torch.empty((n, k, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last) if pool_ok else None
            rc = _lib.load().pp_pw_cat_f16(_ptr(t), _ptr(xx), _ptr(self._wcat), _ptr(self.c3.bias), None, _ptr(y), _ptr(pooled), n * h * w,
                                           h * w, w if pool_ok else 0, c1, c2, k, k, 0, LEAK if self.c3.act else 1.0, _stream(x))
            if rc != 0:
                return None
            return y, (pooled if pool_ok else (maxpool2(y) if want_pool else None))

        choice = _conv_choice.get(key)
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            separate()
            if fused() is None:
                choice = 0
            else:
                t_sep, t_fused = _timed(separate), _timed(fused)
                _conv_timing[key] = {"separate": t_sep, "fused": t_fused}
                choice = 1 if t_fused < t_sep else 0
            _conv_choice[key] = choice
            _note(key, choice)
        return fused() if choice else None


class FHourglass(nn.Module):
    def __init__(self, hg):
        super().__init__()
        self.depth = hg.depth
        self.levels = nn.ModuleList()
        for i in range(hg.depth):
            mods = [FResidual(hg.hg[i][0]), FResidual(hg.hg[i][1]), FResidual(hg.hg[i][2]), _fconv_from_block(hg.hg[i][3])]
            if i == hg.depth - 1:
                mods.append(FResidual(hg.hg[i][4]))
            self.levels.append(nn.ModuleList(mods))

    def _level(self, i, x, coarse, cache0=None, pooled=None):
        """pooled: maxpool2(x) when the producer of x already made it (pp_pw_pool_f16), else None"""
        lv = self.levels[i]
        up1 = lv[0](x)
        pooled = maxpool2(x) if pooled is None else pooled
        if i == self.depth - 1:
            low = lv[4](lv[1](pooled))
        else:   # the next level pools this block's output first: let the block hand it over
            low, low_pooled = lv[1].forward_pool(pooled)
            low = self._level(i + 1, low, coarse, None, low_pooled)
        coarse.append(low)
        # up1 + act(conv(upsample(low)) + b) (+ cache at the top level of stages 2..): one launch when the halo kernel takes it
        return lv[3].forward_up2(lv[2](low), up1, cache0)

    def forward(self, x, cache0=None, pooled=None):
        """-> [top (+ cache0 when given), coarse levels...]"""
        coarse = []
        top = self._level(0, x, coarse, cache0, pooled)
        return [top] + coarse[::-1]


class FSE(nn.Module):
    def __init__(self, se):
        super().__init__()
        self.fc1, self.fc2 = se.fc[0], se.fc[2]

    def gains(self, x, mean=None):
        """sigmoid(fc2(leaky(fc1(mean)))) (n, c): one launch of pp_se_gains_f16 on the GPU (USE_SE_KERNEL), the torch modules elsewhere.
        mean: None, the (n, c) mean, or the producer's partial channel sums (ws, splits, hw) -- see FConv.forward_mean."""
        n, c = x.shape[:2]
        if (USE_SE_KERNEL and x.is_cuda and x.dtype == torch.float16 and self.fc1.weight.dtype == torch.float16
                and self.fc1.bias is not None and self.fc2.bias is not None):
            from . import _lib
            out = torch.empty((n, c), dtype=x.dtype, device=x.device)
            hid = self.fc1.weight.shape[0]
            if isinstance(mean, tuple):
                ws, splits, hw = mean
                rc = _lib.load().pp_se_gains_f16(_ptr(ws), None, _ptr(self.fc1.weight), _ptr(self.fc1.bias), _ptr(self.fc2.weight),
                                                 _ptr(self.fc2.bias), _ptr(out), n, hw, c, hid, splits, 0.01, _stream(x))
            else:
                m = (channel_mean(x) if mean is None else mean).contiguous()
                rc = _lib.load().pp_se_gains_f16(None, _ptr(m), _ptr(self.fc1.weight), _ptr(self.fc1.bias), _ptr(self.fc2.weight),
                                                 _ptr(self.fc2.bias), _ptr(out), n, 0, c, hid, 0, 0.01, _stream(x))
            _lib.check(rc)
            return out
        if isinstance(mean, tuple):
            ws, splits, hw = mean
            mean = (ws.sum(dim=1) / hw).to(x.dtype)
        y = channel_mean(x) if mean is None else mean
        return torch.sigmoid(self.fc2(F.leaky_relu(self.fc1(y), 0.01)))

    def forward(self, x, fold: bool = False, mean=None):
        """mean: the channel mean of x (or the partial sums behind it) when its producer already made it (FConv.forward_mean)"""
        y = self.gains(x, mean)
        if fold and x.is_cuda and x.dtype == torch.float16 and USE_PW and USE_OWN_CONV:
            return Scaled(x, y)            # the consumers (1x1 head / merge convolutions) multiply while they read
        return channel_scale(x, y)


class FFeature(nn.Module):
    def __init__(self, seq):
        super().__init__()
        self.c1, self.c2, self.se = _fconv_from_block(seq[0]), _fconv_from_block(seq[1]), FSE(seq[2])

    def forward(self, x, fold: bool = False):
        y, mean = self.c2.forward_mean(self.c1(x), partial=True)
        return self.se(y, fold, mean)


class FusedIMHN(nn.Module):
    """forward(NHWC image batch in [0,1]) -> (N, 50, H/4, W/4): the `[-1][0]` output of NetworkEval."""

    def __init__(self, net):
        super().__init__()
        p = net.posenet
        self.S, self.K = p.num_stages, p.num_scales
        pre = p.pre
        self.stem = FStem(pre.conv1, pre.bn1, True)
        self.res1, self.res2 = FResidual(pre.res1), FResidual(pre.res2)
        self.dil = nn.ModuleList([_fconv_from_block(d) for d in pre.dilation])
        self.hg = nn.ModuleList([FHourglass(h) for h in p.hourglass])
        self.feat = nn.ModuleList([nn.ModuleList([FFeature(s) for s in f.before_regress]) for f in p.features])
        self.head = nn.ModuleList([nn.ModuleList([FHead(c.conv, c.bn, c.relu is not None) for c in o]) for o in p.outs])
        self.mfeat = nn.ModuleList([nn.ModuleList([_fconv_from_block(m.conv) for m in ms]) for ms in p.merge_features])
        self.mpred = nn.ModuleList([nn.ModuleList([_fconv_from_block(m.conv) for m in ms]) for ms in p.merge_preds])
        # cache_s = merge_pred(pred) + merge_feat(feat): both are bias-only 1x1 convs -> one epilogue with the summed bias
        self.folded_merge = bool(USE_FOLDED_MERGE)
        for t_, (mf, mp) in enumerate(zip(self.mfeat, self.mpred)):
            for s_, (f, q) in enumerate(zip(mf, mp)):
                f.bias = nn.Parameter(f.bias + q.bias, requires_grad=False)
                if self.folded_merge:
                    # cache = merge_features(feat) + merge_preds(head(feat))  (models/posenet.py:116-117) and both merge_preds and
                    # the head are 1x1 convolutions without an activation: merge_preds(head(f)) = (Wp Wh) f + Wp bh is LINEAR in f,
                    # so it folds into merge_features' weights once, like BatchNorm does:  W' = Wf + Wp Wh,  b' = bf + bp + Wp bh.
                    # The prediction-merge convolution, its 256-channel output and the residual read of the merge disappear, and
                    # the heads of the first three stages feed nothing any more (the reference needs them for its training loss).
                    hd = self.head[t_][s_]
                    wp = q.weight.detach().float().flatten(1)                  # (K, 50)
                    wh = hd.weight.detach().float().flatten(1)                 # (50, C)
                    f.weight = nn.Parameter((f.weight.detach().float().flatten(1) + wp @ wh).view_as(f.weight), requires_grad=False)
                    f.bias = nn.Parameter(f.bias.detach().float() + wp @ hd.bias.detach().float(), requires_grad=False)
                # the heads hand over a 64-channel tensor (50 + 14 exact zeros): the prediction-merge convolution reads it
                # as it is, through a weight whose input channels 50..63 are zero (no copy of the [:, :50] view)
                wq = torch.zeros((q.weight.shape[0], 64, 1, 1), dtype=q.weight.dtype)
                wq[:, : q.weight.shape[1]] = q.weight.detach()
                q.w_in64 = nn.Parameter(wq, requires_grad=False)

    @classmethod
    def from_network(cls, net):
        return cls(net)

    def forward(self, imgs, stage_preds: bool = False):
        """stage_preds=True (tests): -> the scale-0 prediction of EVERY stage, i.e. the reference's out[t][0] for t = 0..S-1."""
        seen = []
        x = imgs.permute(0, 3, 1, 2)  # NHWC storage viewed as NCHW == channels_last: no copy
        x = self.stem(x)
        # torch.cat([x, dilation(x)]) of Backbone.forward (models/layers_transposed.py:193-195) without the copy: both halves are
        # written in place -- res2's last 1x1 into [:, :C], the last dilated convolution into [:, C:] -- and the dilated chain reads
        # the first half where it lies (kernels with pixel strides: pp_pw_f16 / pp_conv_ld_f16)
        pooled = self.res1.forward_pool(x)[1]
        if USE_SLICE_OUTPUT and pooled.is_cuda and pooled.dtype == torch.float16:
            c2 = self.res2.c3.weight.shape[0]
            cd = self.dil[-1].weight.shape[0]
            n, _, h, w = pooled.shape
            buf = torch.empty((n, c2 + cd, h, w), dtype=pooled.dtype, device=pooled.device, memory_format=torch.channels_last)
            d = self.res2(pooled, out=buf[:, :c2])
            for i, m in enumerate(self.dil):
                d = m(d, out=buf[:, c2:] if i == len(self.dil) - 1 else None)
            x = buf
        else:
            x = self.res2(pooled)
            d = x
            for m in self.dil:
                d = m(d)
            x = torch.cat([x, d], dim=1)
        caches, x_pooled = None, None
        for t in range(self.S):
            last = t == self.S - 1
            scales = range(1) if last else range(self.K)  # the last stage's coarse heads feed nothing
            hg = self.hg[t](x, None if caches is None else caches[0], x_pooled)   # scale 0 comes back with its cache added
            if caches is not None:
                hg = [hg[0]] + [hg[s] + caches[s] for s in scales if s > 0]
            # feats[s] stays a (feature map, SE gains) pair: its consumers -- the head and the merge convolution, both 1x1 --
            # multiply while they read
            feats = [self.feat[t][s](hg[s], fold=True) for s in scales]
            if last or stage_preds or not self.folded_merge:
                preds = [self.head[t][s](feats[s]) for s in (scales if not self.folded_merge else range(1))]
                seen.append(preds[0])
            if last:
                return seen if stage_preds else preds[0]
            if self.folded_merge:   # merge_preds(head(.)) lives inside merge_features' weights: no head, no prediction merge
                c0, x, x_pooled = self.mfeat[t][0].forward_dual(feats[0], None, x, want_pool=True)
                caches = [c0] + [self.mfeat[t][s](feats[s]) for s in scales if s > 0]
                continue
            # cache_s = merge_feat(feat_s) + merge_pred(pred_s); x + cache_0 leaves the scale-0 convolution as a second output
            mp = [self._merge_pred(t, s, preds[s]) for s in scales]
            c0, x, x_pooled = self.mfeat[t][0].forward_dual(feats[0], mp[0], x, want_pool=True)
            caches = [c0] + [self.mfeat[t][s](feats[s], mp[s]) for s in scales if s > 0]

    def _merge_pred(self, t, s, pred):
        """merge_preds[t][s](pred) without its bias (summed into the feature-merge convolution's)"""
        q, padded = self.mpred[t][s], self.head[t][s].last_padded
        if padded is not None and padded.shape[1] == 64:
            return F.conv2d(padded, q.w_in64)
        return q.conv_only(pred)


class GraphedForward:
    """Replays module(x) from a HIP graph for a fixed input shape (launch-bound coarse levels: ~900 kernels)."""

    def __init__(self, module, example: torch.Tensor, warmup: int = 3):
        self.module = module
        self.static_in = example.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):
                self.module(self.static_in)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = self.module(self.static_in)

    def __call__(self, x):
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out


def build_inference_model(device, fused: bool = True, seed: int = 7, dtype=torch.float16):
    """Random-init (deterministic, name-seeded) IMHN ready for inference on `device`."""
    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from .model_init import deterministic_init
    net = NetworkEval(TrainingOpt(), GetConfig("Canonical"), bn=True).eval()
    deterministic_init(net, seed)
    if not fused:
        return net.to(device=device, dtype=dtype).to(memory_format=torch.channels_last)
    m = FusedIMHN.from_network(net).eval()
    return m.to(device=device, dtype=dtype).to(memory_format=torch.channels_last)
