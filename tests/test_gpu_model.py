"""GPU: the inference form of the IMHN (BN folded, fp16 channels-last, HIP epilogue/pool/upsample kernels)
against the checkpoint-compatible fp32 module, and the helper kernels against plain PyTorch ops."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_helper_kernels_match_torch():
    from posepaf import fused_model as fm
    g = torch.Generator(device="cpu").manual_seed(0)
    for (n, c, h, w) in [(2, 64, 16, 24), (1, 256, 8, 8), (3, 8, 2, 6)]:
        x = torch.randn(n, c, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        r = torch.randn(n, c, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        p = torch.randn(n, c, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        b = torch.randn(c, generator=g).cuda().half()
        for res, act, post in [(None, True, None), (r, True, None), (r, False, None), (None, False, None), (None, True, p), (r, True, p)]:
            y = fm.hip_bias_act_(x.clone(memory_format=torch.channels_last), b, res, act, post)
            ref = x.float() + b.float().view(1, -1, 1, 1)
            if res is not None:
                ref = ref + res.float()
            if act:
                ref = torch.nn.functional.leaky_relu(ref, 0.01)
            if post is not None:
                ref = ref + post.float()
            assert torch.equal(y, ref.half()), (n, c, h, w, res is not None, act, post is not None)
        assert torch.allclose(fm.add3(x, r, p).float(), x.float() + r.float() + p.float(), rtol=0, atol=4e-3)
        assert torch.equal(fm.add3(x, r), x + r)
        sc = torch.rand(n, c, generator=g).cuda().half()
        assert torch.equal(fm.channel_scale(x, sc), x * sc[:, :, None, None])
        m = fm.channel_mean(x)
        assert m.shape == (n, c) and torch.allclose(m.float(), x.float().mean(dim=(2, 3)), rtol=0, atol=2e-3)
        assert torch.equal(fm.maxpool2(x), torch.nn.functional.max_pool2d(x, 2, 2))
        assert torch.equal(fm.upsample2(x), torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"))


def test_fused_fp16_forward_close_to_fp32_module():
    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from posepaf.fused_model import FusedIMHN
    from posepaf.model_init import deterministic_init
    net = NetworkEval(TrainingOpt(), GetConfig("Canonical"), bn=True).eval()
    deterministic_init(net, 7)
    x = torch.from_numpy(np.random.default_rng(3).random((2, 128, 192, 3), dtype=np.float32))
    with torch.no_grad():
        want = net.cuda()(x.cuda())[-1][0].float().cpu()
        fused = FusedIMHN.from_network(net).eval().cuda().half().to(memory_format=torch.channels_last)
        got = fused(x.cuda().half()).float().cpu()
    assert got.shape == want.shape == (2, 50, 32, 48)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err < 0.03 * scale, (err, scale)   # fp16 activations through ~300 convolutions
    assert torch.isfinite(got).all()


def test_fused_fp16_forward_at_the_bench_geometry_stage_by_stage():
    """The fused fp16 forward at 512 x 512 (two images: the 128-wide halo tiles, the upsample / two-output fusions at their
    real sizes, every kernel the tuner picks there) against the fp32 module, STAGE BY STAGE: the scale-0 prediction of stage t
    is out[t][0] of the reference module (models/posenet.py:90-122).  Budget per stage, relative to that stage's output
    scale: fp16 storage of ~75 convolution outputs per stage, errors of earlier stages feed the later ones through the
    caches -- measured on MI355X (round 3): max 0.22 / 0.25 / 0.34 / 0.34 %, rms 0.043 / 0.043 / 0.072 / 0.069 % for stages
    1..4; the budget is about twice that.  The old bound (3 % of the last stage's max at 128 x 192) stays above as the coarse
    check."""
    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from posepaf.fused_model import FusedIMHN
    from posepaf.model_init import deterministic_init
    net = NetworkEval(TrainingOpt(), GetConfig("Canonical"), bn=True).eval()
    deterministic_init(net, 7)
    x = torch.from_numpy(np.random.default_rng(5).random((2, 512, 512, 3), dtype=np.float32))
    with torch.no_grad():
        ref = [st[0].float().cpu() for st in net.cuda()(x.cuda())]
        fused = FusedIMHN.from_network(net).eval().cuda().half().to(memory_format=torch.channels_last)
        fused(x.cuda().half())                                   # tunes the shapes of this geometry
        got = [g.float().cpu() for g in fused(x.cuda().half(), stage_preds=True)]
    assert len(got) == len(ref) == 4
    budget_max = [0.005, 0.006, 0.008, 0.008]
    budget_rms = [0.0010, 0.0010, 0.0015, 0.0015]
    report = []
    for t, (g, w) in enumerate(zip(got, ref)):
        assert g.shape == w.shape == (2, 50, 128, 128) and torch.isfinite(g).all()
        scale = w.abs().max().item()
        e_max = (g - w).abs().max().item() / scale
        e_rms = (g - w).pow(2).mean().sqrt().item() / scale
        report.append((t, round(e_max, 5), round(e_rms, 6)))
    print("stage errors (max, rms) / stage scale:", report)
    for t, e_max, e_rms in report:
        assert e_max <= budget_max[t] and e_rms <= budget_rms[t], report


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 128, 448), (3, 512, 512), (1, 192, 320)])
def test_stem_kernel_matches_torch(shape):
    """pp_stem7x7_f16 (Backbone.conv1 + bn1 + LeakyReLU, models/layers_transposed.py:78-87) against the fp32 torch convolution
    of the same fp16 operands: image borders (zero padding of the convolution on all four sides), widths whose half is not a
    multiple of the 64-column tile (448 -> 224, 320 -> 160), several tiles per workgroup; <= 2e-3 of the output scale."""
    import torch.nn.functional as F
    from posepaf.fused_model import FStem
    n, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(31)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = torch.nn.BatchNorm2d(64).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / 147 ** 0.5)
        bn.weight.copy_(0.5 + torch.rand(64, generator=g)), bn.bias.copy_(0.2 * torch.randn(64, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(64, generator=g)), bn.running_var.copy_(0.5 + torch.rand(64, generator=g))
    stem = FStem(conv, bn, True).cuda().half()
    x = torch.rand((n, h, w, 3), generator=g).cuda().half()
    with torch.no_grad():
        got = stem(x.permute(0, 3, 1, 2)).float()
        ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2).float(), stem.weight.float(), stem.bias.float(), 2, 3), 0.01)
    assert got.shape == ref.shape == (n, 64, h // 2, w // 2) and torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (shape, err)


@pytest.mark.parametrize("shape", [
    # n, h, w, c_in, c_out
    (2, 16, 32, 128, 256),      # weights 64 KB: two workgroups per CU
    (3, 8, 8, 256, 256),        # 128 KB of weights: one workgroup per CU; 192 pixels = 6 groups
    (1, 8, 12, 256, 640),       # C_out split over five workgroup columns (640 x 256 x 2 B > LDS)
    (2, 5, 7, 64, 64),          # 70 pixels: a ragged last group (no scale: hw % 64 != 0); narrow input: 64-pixel groups
    (1, 16, 16, 384, 192),      # 144 KB of weights
    (2, 8, 16, 512, 128),
    (1, 32, 32, 192, 384),
    (2, 4, 64, 64, 128),        # narrow input (64-pixel groups) with a row width the pooled form takes
])
def test_streaming_pointwise_convolution_matches_torch(shape):
    """pp_pw_f16 (csrc/posepaf_conv_own.hip k_pw): y = act(conv1x1(x * scale[n]) + bias (+ extra)) [, y2 = y + extra2] against fp32
    torch on the same fp16 operands -- every epilogue mode, the SE gains folded into the input read (must equal the convolution
    of the fp16-rounded x * s tensor), a strided output (channel slice of a wider tensor), C_out splits, ragged pixel counts."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib
    L = _lib.load()
    n, h, w, ci, co = shape
    assert L.pp_pw_supported(ci, co)
    g = torch.Generator(device="cpu").manual_seed(41)
    x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    ex = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    ex2 = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    sc = (0.2 + torch.rand(n, ci, generator=g)).cuda().half()
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    m, hw = n * h * w, h * w
    for use_scale in ([False, True] if hw % 64 == 0 else [False]):
        xin = (x * sc[:, :, None, None]) if use_scale else x          # the fp16 tensor the separate SE pass would have written
        conv = F.conv2d(xin.float(), wt.float(), b.float())
        for mode, slope in [(0, 0.01), (1, 0.01), (2, 0.01), (0, 1.0), (4, 1.0), (5, 0.01)]:
            ref = conv + ex.float() if mode in (1, 4) else conv
            ref = F.leaky_relu(ref, slope) if slope != 1.0 else ref
            ref = ref + ex.float() if mode == 2 else ref
            wide = torch.full((n, co + 64, h, w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            y = wide[:, 32:32 + co]                                    # a channel slice: pixel stride co + 64
            y2 = torch.full((n, co, h, w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            rc = L.pp_pw_f16(vp(x.data_ptr()), vp(sc.data_ptr()) if use_scale else None, vp(wt.data_ptr()), vp(b.data_ptr()),
                             vp(ex.data_ptr()) if mode in (1, 2, 4) else None, vp(ex2.data_ptr()) if mode >= 4 else None, vp(y.data_ptr()),
                             vp(y2.data_ptr()) if mode >= 4 else None, m, hw, ci, co, co + 64, mode, slope, st)
            assert rc == 0, (shape, mode, rc)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and torch.isnan(wide[:, :32]).all() and torch.isnan(wide[:, 32 + co:]).all()
            err = (y.float() - ref).abs().max().item()
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (shape, use_scale, mode, err)
            if mode >= 4:
                assert torch.equal(y2, (y.float() + ex2.float()).half())   # the exact sum of the two binary16 tensors, rounded once
    assert L.pp_pw_f16(vp(x.data_ptr()), None, vp(wt.data_ptr()), vp(b.data_ptr()), None, None, vp(x.data_ptr()), None, m, hw, 96, co,
                       co, 0, 0.01, st) == -6                            # an input width the kernel has no instance for
    # the pooled second output (pp_pw_pool_f16): groups become 2-row blocks; y must not change by a bit and the pooled tensor
    # must be exactly the 2x2 max-pool of it (of y2 in mode 4)
    if h % 2 == 0 and w % (64 if ci == 64 else 32) == 0:
        for mode in (1, 4):
            ys, y2s = [], []
            for pool in (False, True):
                y = torch.zeros((n, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
                y2 = torch.zeros_like(y, memory_format=torch.channels_last)
                pooled = torch.full((n, co, h // 2, w // 2), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
                common = (vp(x.data_ptr()), None, vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()), vp(ex2.data_ptr()) if mode == 4 else None,
                          vp(y.data_ptr()), vp(y2.data_ptr()) if mode == 4 else None)
                if pool:
                    rc = L.pp_pw_pool_f16(*common, vp(pooled.data_ptr()), m, hw, w, ci, co, co, mode, 0.01, st)
                else:
                    rc = L.pp_pw_f16(*common, m, hw, ci, co, co, mode, 0.01, st)
                assert rc == 0, (shape, mode, pool, rc)
                torch.cuda.synchronize()
                ys.append(y), y2s.append(y2)
            assert torch.equal(ys[0], ys[1]) and torch.equal(y2s[0], y2s[1])
            assert torch.equal(pooled, F.max_pool2d(y2s[1] if mode == 4 else ys[1], 2, 2))


@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 32), (1, 256, 256, 16, 64), (3, 96, 192, 16, 128)])
def test_halo_kernel_emits_the_channel_sums_of_its_output(shape):
    """pp_conv_own_sums_f16 + pp_channel_mean_finish_f16: y bit-equal to the plain halo-kernel launch, and the mean equal to the
    fp32 mean of that fp16 tensor within fp32 summation-order noise (the SE squeeze, models/layers_transposed.py:298-303)."""
    import ctypes as C
    from posepaf import _lib
    L = _lib.load()
    n, ci, co, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(53)
    x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    splits = L.pp_conv_own_sums_splits(h, w)
    assert splits > 0
    y0 = torch.empty((n, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    assert L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y0.data_ptr()), n, h, w, ci, co, 3, 1, 1, 0,
                             0.01, 512, st) == 0
    y1 = torch.empty_like(y0, memory_format=torch.channels_last)
    ws = torch.full((n, splits, co), float("nan"), dtype=torch.float32, device="cuda")
    mean = torch.empty((n, co), dtype=torch.float16, device="cuda")
    assert L.pp_conv_own_sums_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(y1.data_ptr()), vp(ws.data_ptr()), n, h, w, ci,
                                  co, 0.01, st) == 0
    assert L.pp_channel_mean_finish_f16(vp(ws.data_ptr()), vp(mean.data_ptr()), n, h * w, co, splits, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.isfinite(ws).all()
    want = y1.float().mean(dim=(2, 3))
    assert (mean.float() - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 8, 64, 64, 64, 128), (1, 16, 32, 128, 64, 256), (3, 6, 10, 64, 192, 128),
                                   (2, 16, 32, 192, 256, 384), (2, 8, 32, 256, 384, 512), (1, 16, 16, 320, 384, 640)])
def test_two_input_pointwise_convolution_matches_the_sum_of_two(shape):
    """pp_pw_cat_f16: act(W [x ; x2] + b) == act(conv1x1(x, W[:, :c1]) + conv1x1(x2, W[:, c1:]) + b) -- a residual block's last 1x1 and
    its skip convolution as one product (models/layers_transposed.py:12-48) -- incl. the pooled output."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib
    L = _lib.load()
    n, h, w, c1, c2, co = shape
    g = torch.Generator(device="cpu").manual_seed(71)
    x1 = torch.randn(n, c1, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    x2 = torch.randn(n, c2, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, c1 + c2, generator=g) / (c1 + c2) ** 0.5).cuda().half().contiguous()
    b = torch.randn(co, generator=g).cuda().half()
    ref = F.leaky_relu(F.conv2d(x1.float(), wt[:, :c1].float()[:, :, None, None]) + F.conv2d(x2.float(), wt[:, c1:].float()[:, :, None, None])
                       + b.float()[None, :, None, None], 0.01)
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    pool_ok = h % 2 == 0 and w % (64 if c1 + c2 == 64 else 32) == 0 and c1 + c2 <= 512   # (16-pixel groups beyond: no pooled output)
    y = torch.full((n, co, h, w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    pooled = torch.full((n, co, h // 2, w // 2), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    rc = L.pp_pw_cat_f16(vp(x1.data_ptr()), vp(x2.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()),
                         vp(pooled.data_ptr()) if pool_ok else None, n * h * w, h * w, w if pool_ok else 0, c1, c2, co, co, 0, 0.01, st)
    assert rc == 0, (shape, rc)
    torch.cuda.synchronize()
    assert (y.float() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    if pool_ok:
        assert torch.equal(pooled, F.max_pool2d(y, 2, 2))


def test_folded_prediction_merge_keeps_every_stage(monkeypatch=None):
    """merge_preds(head(f)) folded into merge_features' weights (W' = Wf + Wp Wh, b' = bf + bp + Wp bh; models/posenet.py:116-117)
    against the unfolded model (heads + prediction-merge convolutions computed at every stage): every stage's prediction within
    fp16 noise of the other form (the unfolded form rounds the intermediate predictions to binary16, the folded one does not)."""
    from posepaf import fused_model as fm
    x = torch.from_numpy(np.random.default_rng(13).random((2, 128, 128, 3), dtype=np.float32)).cuda().half()
    outs = []
    for folded in (True, False):
        fm.USE_FOLDED_MERGE = folded
        try:
            model = fm.build_inference_model(torch.device("cuda"), fused=True)
        finally:
            fm.USE_FOLDED_MERGE = True
        assert model.folded_merge == folded
        with torch.no_grad():
            outs.append([o.float() for o in model(x, stage_preds=True)])
    for t, (a, b) in enumerate(zip(*outs)):
        assert a.shape == b.shape == (2, 50, 32, 32)
        assert (a - b).abs().max().item() <= 0.01 * b.abs().max().item(), t


def test_se_gains_folded_into_the_consumers_keep_the_model_output():
    """FusedIMHN with the SE gains folded into the head / merge convolutions' input read (Scaled + pp_pw_f16) against the same
    model with the separate x * s pass (USE_PW = False): same arithmetic up to the accumulation order of the 1x1 kernels."""
    from posepaf import fused_model as fm
    model = fm.build_inference_model(torch.device("cuda"), fused=True)
    x = torch.from_numpy(np.random.default_rng(9).random((2, 128, 128, 3), dtype=np.float32)).cuda().half()
    with torch.no_grad():
        fm.USE_PW = False
        try:
            a = model(x).float()
        finally:
            fm.USE_PW = True
        b = model(x).float()
    assert a.shape == b.shape == (2, 50, 32, 32)
    assert (a - b).abs().max().item() <= 0.01 * a.abs().max().item()


def test_pipeline_end_to_end_runs():
    """uint8 images -> records through the real architecture.  (Bitwise run-to-run equality is NOT asserted here:
    MIOpen may pick a different convolution algorithm on a shape's first call and some of its fp16 kernels
    accumulate with atomics, so a random-weight network's noise-level peaks can differ between calls; the
    post-processing kernels themselves are checked for determinism in test_gpu_parity.py.)"""
    from posepaf.api import PosePostProcessor
    from posepaf.fused_model import build_inference_model
    from posepaf.pipeline import PosePipeline
    post = PosePostProcessor(max_batch=2, max_h=64, max_w=64)
    model = build_inference_model(torch.device("cuda", 0))
    pipe = PosePipeline(model, post)
    img = torch.randint(0, 256, (2, 256, 256, 3), dtype=torch.uint8, device="cuda")
    maps = pipe.forward_maps(img)
    assert maps.shape == (2, 2, 50, 64, 64) and maps.dtype == torch.float16 and torch.isfinite(maps).all()
    rec = post.process(maps, 256)
    again = post.process(maps, 256)
    assert rec.shape == (2,) and rec.tobytes() == again.tobytes()          # same maps -> identical records
    assert ((rec["n_humans"] >= 0) & (rec["n_humans"] <= 128)).all()
    r = pipe(img)
    assert r.shape == (2,)


@pytest.mark.parametrize("shape", [(37, 70), (512, 512), (480, 640)])
def test_preprocess_kernel_matches_host_formula(shape):
    """pp_preprocess_u8 (A0) == the reference's pad / 255 / flip expression, fp32 exactly and fp16 after rounding."""
    from posepaf.pipeline import preprocess_batch
    h, w = shape
    img = torch.from_numpy(np.random.default_rng(h).integers(0, 256, (3, h, w, 3), dtype=np.uint8))
    for dt in (torch.float32, torch.float16):
        want = preprocess_batch(img, True, dt)
        got = preprocess_batch(img.cuda(), True, dt).cpu()
        assert got.shape == want.shape and torch.equal(got, want)
        assert torch.equal(preprocess_batch(img.cuda(), False, dt).cpu(), preprocess_batch(img, False, dt))


@pytest.mark.parametrize("K,N", [(64, 64), (128, 256), (256, 128), (192, 96), (128, 128), (128, 32)])
def test_pwconv_mfma_kernel_matches_torch(K, N):
    """pp_pwconv_f16 (fused 1x1 conv on v_mfma_f32_32x32x16_f16) vs fp32 math on the same fp16 operands.
    ASYMMETRIC operands (random) and an M that is not a multiple of the 128-row tile."""
    import ctypes as C
    from posepaf import _lib
    L = _lib.load()
    assert L.pp_pwconv_supported(K, N) == 1
    g = torch.Generator(device="cpu").manual_seed(K * 1000 + N)
    M = 128 * 5 + 37
    x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda()
    p = torch.randn(M, N, generator=g).half().cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for res, act, post in [(None, 0, None), (None, 1, None), (r, 1, None), (r, 1, p), (None, 0, p)]:
        y = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        rc = L.pp_pwconv_f16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                             C.c_void_p(res.data_ptr()) if res is not None else None,
                             C.c_void_p(post.data_ptr()) if post is not None else None, C.c_void_p(y.data_ptr()), M, K, N, 0.01, act, st)
        assert rc == 0
        ref = (x.float() @ w.float().t() + b.float()).half().float()       # kernel rounds conv + bias to fp16 once
        if res is not None:
            ref = ref + res.float()
        if act:
            ref = torch.nn.functional.leaky_relu(ref, 0.01)
        if post is not None:
            ref = ref + post.float()
        err = (y.float() - ref).abs().max().item()
        assert torch.isfinite(y).all()
        assert err < 2e-2, (K, N, res is not None, act, post is not None, err)
    assert L.pp_pwconv_supported(384, 192) == 0 and L.pp_pwconv_supported(256, 384) == 0 and L.pp_pwconv_supported(256, 256) == 0


@pytest.mark.parametrize("shape", [
    # n, c_in, c_out, h, w, ksize, pad, dilation
    (2, 64, 64, 32, 32, 3, 1, 1),
    (1, 128, 128, 24, 40, 3, 3, 3),     # the dilated 3x3 of the feature heads
    (2, 256, 128, 16, 16, 1, 0, 1),
    (1, 192, 384, 8, 8, 1, 0, 1),
    (3, 72, 40, 9, 7, 3, 1, 1),          # ragged: nothing is a multiple of a tile
])
def test_fused_convolution_matches_torch(shape):
    """pp_conv_f16 (every tile configuration that accepts the shape) against an fp32 torch convolution of the same
    fp16 operands: bias, LeakyReLU, residual-before-activation and add-after-activation variants.  fp32 accumulate, fp16
    store: tolerance 2e-3 relative to the output scale."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib
    L = _lib.load()
    n, ci, co, h, w, k, pad, dil = shape
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    ho, wo = h + 2 * pad - dil * (k - 1), w + 2 * pad - dil * (k - 1)
    ex = torch.randn(n, co, ho, wo, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    conv = F.conv2d(x.float(), wt.float(), b.float(), 1, pad, dil)
    vp = C.c_void_p
    stream = vp(torch.cuda.current_stream().cuda_stream)
    ran = 0
    for mode, slope in [(0, 0.01), (1, 0.01), (2, 0.01), (0, 1.0), (1, 1.0)]:
        ref = conv + ex.float() if mode == 1 else conv
        ref = F.leaky_relu(ref, slope) if slope != 1.0 else ref
        ref = ref + ex.float() if mode == 2 else ref
        for cfg in range(L.pp_conv_num_configs()):
            y = torch.full((n, co, ho, wo), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            rc = L.pp_conv_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None,
                               vp(y.data_ptr()), n, h, w, ci, co, k, pad, dil, mode, slope, cfg, stream)
            if rc == -6:   # PP_ERR_UNSUPPORTED: this tile configuration does not take the shape
                continue
            assert rc == 0, (cfg, mode, rc)
            torch.cuda.synchronize()
            err = (y.float() - ref).abs().max().item()
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (shape, cfg, mode, slope, err)
            ran += 1
    assert ran >= 5, "no tile configuration accepted this shape"
    # argument errors are loud
    assert L.pp_conv_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), n, h, w, ci, co, k, pad,
                         dil, 1, 0.01, 0, stream) == -2          # extra_mode without a tensor
    assert L.pp_conv_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), n, h, w, ci, co, k, pad,
                         dil, 0, 0.01, 99, stream) == -2         # unknown configuration


def test_fused_convolution_autotune_keeps_model_output():
    """The forward with the fused convolutions on and off (MIOpen + epilogue pass) agrees within fp16 noise."""
    from posepaf import fused_model as fm
    model = fm.build_inference_model(torch.device("cuda", 0))
    x = torch.from_numpy(np.random.default_rng(4).random((2, 128, 128, 3), dtype=np.float32)).cuda().half()
    with torch.no_grad():
        fm.USE_FUSED_CONV = False
        try:
            a = model(x).float()
        finally:
            fm.USE_FUSED_CONV = True
        b = model(x).float()
    assert any(v >= 0 for v in fm.conv_choices().values()), "no layer chose a fused configuration"
    assert (a - b).abs().max().item() <= 0.02 * a.abs().max().item()


def test_evaluate_script_with_two_image_sizes(tmp_path):
    """improved-body-parts_amd/evaluate.py end to end on a heterogeneous synthetic set (two image sizes whose padded shapes
    differ, so two buckets; batch 2 with a ragged last batch), both rule sets: records come back in image order, nothing
    overflows, and the injected people are recovered (in-repo OKS AP; pycocotools is absent -> parity unpinned)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG
    for rules in (["--run_cpp"], []):
        dump = tmp_path / "res.json"
        r = subprocess.run([sys.executable, os.path.join(PKG, "evaluate.py"), "--run_refactor", *rules, "--synthetic", "7",
                            "--sizes", "256x256,200x300", "--batch", "2", "--people", "2", "3", "--dump_name", str(dump)],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        summary = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert summary["images"] == 7 and summary["status_or"] == 0
        res = json.load(open(dump))
        assert summary["people_found"] == len(res) >= 7 * 2 - 2
        assert sorted({r_["image_id"] for r_ in res}) == list(range(7))
        assert summary["synthetic_oks"]["AP"] > 0.5, summary


@pytest.mark.parametrize("shape", [
    # n, c_in, c_out, h, w, ksize, pad, dilation
    (2, 64, 64, 32, 32, 3, 1, 1),        # BN = 64
    (1, 128, 128, 24, 40, 3, 3, 3),      # the dilated 3x3 of the feature heads, BN = 128
    (2, 256, 256, 16, 24, 3, 1, 1),      # the hot layer's shape family, BN = 256
    (2, 256, 128, 16, 16, 1, 0, 1),      # 1x1
    (1, 192, 384, 8, 8, 1, 0, 1),        # 1x1, BN = 128, three channel tiles
    (3, 64, 192, 9, 7, 3, 1, 1),         # ragged pixel count (189 rows: a partial 256-row tile), BN = 64 x 3
    (1, 320, 640, 5, 6, 3, 1, 1),        # more K-steps than pixels
    (5, 64, 128, 120, 128, 1, 0, 1),     # 300 pixel tiles x 1 / 2 channel tiles on a persistent grid of 256 (512): several tiles per
                                         # workgroup, the last ones on some workgroups only; two phases per tile (short K loop)
    (3, 32, 64, 100, 128, 3, 2, 2),      # 150 ragged pixel tiles, dilated, nine phases per tile, one channel block
])
def test_own_implicit_gemm_convolution_matches_torch(shape):
    """pp_conv_own_f16 (hand-written LDS-DMA + MFMA implicit GEMM, csrc/posepaf_conv_own.hip) against an fp32 torch convolution
    of the same fp16 operands, every epilogue variant and every workgroup-tile width that divides c_out.  ASYMMETRIC random
    operands; fp32 accumulate, fp16 store: tolerance 2e-3 relative to the output scale (same bar as the template kernels)."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib
    L = _lib.load()
    n, ci, co, h, w, k, pad, dil = shape
    assert L.pp_conv_own_supported(ci, co, k) == 1
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    ho, wo = h + 2 * pad - dil * (k - 1), w + 2 * pad - dil * (k - 1)
    ex = torch.randn(n, co, ho, wo, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    conv = F.conv2d(x.float(), wt.float(), b.float(), 1, pad, dil)
    vp = C.c_void_p
    stream = vp(torch.cuda.current_stream().cuda_stream)
    ran = 0
    for mode, slope in [(0, 0.01), (1, 0.01), (2, 0.01), (0, 1.0), (1, 1.0)]:
        ref = conv + ex.float() if mode == 1 else conv
        ref = F.leaky_relu(ref, slope) if slope != 1.0 else ref
        ref = ref + ex.float() if mode == 2 else ref
        for bn in (0, 256, 128, 64):
            y = torch.full((n, co, ho, wo), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            rc = L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None,
                                   vp(y.data_ptr()), n, h, w, ci, co, k, pad, dil, mode, slope, bn, stream)
            if rc == -6:
                assert bn and co % bn      # only a tile width that does not divide c_out may be refused
                continue
            assert rc == 0, (bn, mode, rc)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all(), (shape, bn, mode)
            err = (y.float() - ref).abs().max().item()
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (shape, bn, mode, slope, err)
            ran += 1
    assert ran >= 10
    assert L.pp_conv_own_supported(72, 64, 3) == 0 and L.pp_conv_own_supported(64, 40, 3) == 0


@pytest.mark.parametrize("shape", [
    # n, c_in, c_out, h, w   (3x3, pad 1): tile 512 pixels = (512 / min(w, 128)) rows x min(w, 128) columns
    (2, 64, 128, 32, 32),       # one 16 x 32 tile pair per image, 2 channel blocks
    (1, 256, 256, 16, 64),      # 8 x 64 tiles, two output-channel tiles, 8 channel blocks
    (1, 96, 128, 128, 128),     # 4 x 128 tiles (the hot layer's geometry), 3 channel blocks, 32 tiles
    (2, 32, 384, 8, 256),       # 256-wide image: two 4 x 128 tiles per row pair; one channel block (no halo prefetch)
    (9, 64, 256, 64, 128),      # 288 workgroup ids on a 256-workgroup persistent grid: 32 workgroups walk two tiles
    (131, 32, 128, 32, 32),     # 262 pixel tiles: 33 per XCD range with two padding ids, second tiles on 8 workgroups only
    (2, 64, 128, 16, 192),      # width 192 (the 768-pixel input of the scale search): three 8 x 64 tiles per row band
    (1, 64, 256, 32, 96),       # width 96: three 16 x 32 tiles per row band
    (1, 64, 192, 32, 64),       # C_out = 64 mod 128: the second channel tile is half empty (zero weight rows, epilogue skipped)
    (2, 32, 64, 16, 128),       # C_out = 64: one half-empty channel tile
    (4, 64, 128, 16, 16),       # a 16 x 16 map: two whole images stacked in one 16 x 32 tile, each with its own halo rows
    (6, 32, 320, 16, 16),       # ... with C_out = 64 mod 128 and an odd number of image pairs
])
def test_halo_tile_3x3_convolution_matches_torch(shape):
    """pp_conv_own_f16 with bn = 512: the 3x3 kernel that keeps the input halo of a 512-pixel tile in LDS and reads the nine
    taps from it (csrc/posepaf_conv_own.hip k_conv3x3_halo).  Image borders (zero padding), tile borders (halo from the
    neighbouring tile's pixels), every epilogue variant; asymmetric random operands; tolerance as for the other kernels."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib
    L = _lib.load()
    n, ci, co, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(23)
    x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    ex = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    conv = F.conv2d(x.float(), wt.float(), b.float(), 1, 1, 1)
    vp = C.c_void_p
    stream = vp(torch.cuda.current_stream().cuda_stream)
    for mode, slope in [(0, 0.01), (1, 0.01), (2, 0.01), (0, 1.0)]:
        ref = conv + ex.float() if mode == 1 else conv
        ref = F.leaky_relu(ref, slope) if slope != 1.0 else ref
        ref = ref + ex.float() if mode == 2 else ref
        for rep in range(3):     # the same launch repeatedly: a race between the DMA ring and the fragment reads would not repeat
            variant = 512
            y = torch.full((n, co, h, w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            rc = L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None,
                                   vp(y.data_ptr()), n, h, w, ci, co, 3, 1, 1, mode, slope, variant, stream)
            assert rc == 0, (mode, rc)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all(), (shape, mode, variant)
            err = (y.float() - ref).abs().max().item()
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (shape, mode, slope, rep, variant, err)
    # shapes the halo kernel does not take are refused (bn = 512) and routed to the implicit-GEMM kernel by bn = 0
    y = torch.empty((n, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    assert L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), n, h, w, ci, co, 3, 2, 2,
                             0, 0.01, 512, stream) == -6


@pytest.mark.parametrize("shape", [
    # n, c_in, c_out, h, w, dilation (= padding)
    (2, 128, 128, 128, 128, 3),   # the backbone's geometry at 512 x 512: 4 x 128 tiles, 3 row classes of 43 / 43 / 42 rows (11 tiles each)
    (2, 128, 128, 128, 128, 4),   # 4 classes of 32 rows: every tile row is real
    (1, 128, 128, 128, 128, 5),   # 5 classes of 26 / 26 / 26 / 25 / 25 rows in 7 tiles each
    (1, 64, 128, 50, 64, 3),      # 8 x 64 tiles, classes of 17 / 17 / 16 rows: the last tile of a class holds one or no real row
    (3, 32, 192, 37, 96, 5),      # 16 x 32 tiles, three per row band, classes shorter than a tile; C_out = 64 mod 128
    (1, 64, 64, 7, 32, 4),        # fewer rows than two per class
    (5, 32, 128, 24, 256, 4),     # two 128-wide tiles per row: the halo crosses the tile border by 4 columns
])
def test_dilated_3x3_convolution_on_the_halo_kernel_matches_torch(shape):
    """pp_conv_own_f16 with bn = 512 and dilation 3 / 4 / 5 (models/layers_transposed.py:125-157, the backbone's DilatedConv
    stack): the halo kernel walks the image's rows class by class (y mod d) and reads its taps at column offsets 0, d, 2 d of a
    halo row d pixels wider on each side.  Borders, classes that end inside a tile, every epilogue variant; same tolerance."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib
    L = _lib.load()
    n, ci, co, h, w, d = shape
    g = torch.Generator(device="cpu").manual_seed(29)
    x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    ex = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    conv = F.conv2d(x.float(), wt.float(), b.float(), 1, d, d)
    vp = C.c_void_p
    stream = vp(torch.cuda.current_stream().cuda_stream)
    for mode, slope in [(0, 0.01), (1, 0.01), (2, 0.01), (0, 1.0)]:
        ref = conv + ex.float() if mode == 1 else conv
        ref = F.leaky_relu(ref, slope) if slope != 1.0 else ref
        ref = ref + ex.float() if mode == 2 else ref
        for rep in range(2):
            y = torch.full((n, co, h, w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            rc = L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None,
                                   vp(y.data_ptr()), n, h, w, ci, co, 3, d, d, mode, slope, 512, stream)
            assert rc == 0, (mode, rc)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all(), (shape, mode)     # every pixel written exactly by some tile
            err = (y.float() - ref).abs().max().item()
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (shape, mode, slope, rep, err)
    # padding != dilation is not this kernel's
    y = torch.empty((n, co, h - 2, w - 2), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    assert L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), n, h, w, ci, co, 3, d - 1, d,
                             0, 0.01, 512, stream) == -6


@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 128, 1), (1, 128, 128, 64, 128, 3), (2, 32, 64, 24, 64, 5)])
def test_halo_kernel_reads_and_writes_channel_slices_in_place(shape):
    """pp_conv_own_ld_f16: x and y are channel slices of wider NHWC tensors (the backbone's concatenation,
    models/layers_transposed.py:193-195): same values as the packed call, and not a byte outside the slice is written."""
    import ctypes as C
    from posepaf import _lib
    L = _lib.load()
    n, ci, co, h, w, d = shape
    g = torch.Generator(device="cpu").manual_seed(31)
    xw = torch.randn(n, 40 + ci + 24, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    x = xw[:, 40:40 + ci]
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    ex = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    xp = x.contiguous(memory_format=torch.channels_last)
    for mode in (0, 1):
        want = torch.empty((n, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
        assert L.pp_conv_own_f16(vp(xp.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None,
                                 vp(want.data_ptr()), n, h, w, ci, co, 3, d, d, mode, 0.01, 512, st) == 0
        yw = torch.full((n, 16 + co + 8, h, w), 7.0, dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
        y = yw[:, 16:16 + co]
        assert L.pp_conv_own_ld_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(ex.data_ptr()) if mode else None,
                                    vp(y.data_ptr()), n, h, w, ci, co, 3, d, d, mode, 0.01, 512, x.stride(1) and xw.shape[1],
                                    yw.shape[1], st) == 0
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        assert (yw[:, :16] == 7.0).all() and (yw[:, 16 + co:] == 7.0).all()
    assert L.pp_conv_own_ld_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), n, h, w, ci, co, 3, d, d,
                                0, 0.01, 128, xw.shape[1], yw.shape[1], st) == -6     # the implicit-GEMM kernels read packed pixels


@pytest.mark.parametrize("shape", [
    # n, c_in, c_out, h_low, w_low: the convolution runs at (2 h_low, 2 w_low)
    (2, 64, 128, 16, 16),       # 32-wide tiles
    (1, 128, 256, 8, 64),       # 128-wide tiles, two output-channel tiles
    (3, 32, 128, 32, 32),       # 64-wide tiles, one channel block
    (1, 64, 128, 8, 96),        # 192-wide output (three 64-wide tiles) read from a 96-wide half-resolution input
    (4, 64, 128, 8, 8),         # 16 x 16 output (two images per tile) read from an 8 x 8 half-resolution input
])
def test_halo_kernel_upsampled_input_and_two_post_adds(shape):
    """pp_conv_own_ex_f16: the x2 nearest upsample (models/layers_transposed.py:212, :272) read through the 3x3 kernel's halo loads
    (upsampled_input = 1) must equal the convolution of the materialised upsample BIT FOR BIT (same kernel, same operands in
    LDS), and extra_mode 3 must equal fp16(act) + extra + extra2 summed in fp32 and rounded once (what the three-way add
    kernel does with the fp16 convolution output)."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib
    L = _lib.load()
    n, ci, co, hl, wl = shape
    h, w = 2 * hl, 2 * wl
    g = torch.Generator(device="cpu").manual_seed(29)
    low = torch.randn(n, ci, hl, wl, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    up = F.interpolate(low, scale_factor=2, mode="nearest").contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, generator=g).cuda().half()
    e1 = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    e2 = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    vp = C.c_void_p
    stream = vp(torch.cuda.current_stream().cuda_stream)

    def run(x, mode, upflag, variant=512):
        y = torch.full((n, co, h, w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
        rc = L.pp_conv_own_ex_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(e1.data_ptr()) if mode else None,
                                  vp(e2.data_ptr()) if mode == 3 else None, vp(y.data_ptr()), None, n, h, w, ci, co, 3, 1, 1, mode, 0.01,
                                  variant, upflag, stream)
        torch.cuda.synchronize()
        return rc, y

    for mode in (0, 2, 3):
        rc_a, y_up = run(low, mode, 1)
        rc_b, y_mat = run(up, mode, 0)
        assert rc_a == 0 and rc_b == 0, (mode, rc_a, rc_b)
        assert torch.isfinite(y_up).all()
        assert torch.equal(y_up, y_mat), (shape, mode, (y_up.float() - y_mat.float()).abs().max().item())
    # mode 3 against its definition, built from the mode-0 output of the same kernel
    _, y0 = run(up, 0, 0)
    _, y3 = run(up, 3, 0)
    assert torch.equal(y3, (y0.float() + e1.float() + e2.float()).half())
    # and against torch in fp32, with the tolerance of the other convolution tests
    ref = F.leaky_relu(F.conv2d(up.float(), wt.float(), b.float(), 1, 1, 1), 0.01) + e1.float() + e2.float()
    assert (y3.float() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    # refused where the extension does not exist: other kernels, odd sizes, inconsistent pointers
    assert run(low, 2, 1, variant=256)[0] == -6 and run(low, 3, 0, variant=0)[0] == -6
    y = torch.empty((n, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    assert L.pp_conv_own_ex_f16(vp(low.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(e1.data_ptr()), None, vp(y.data_ptr()), None,
                                n, h, w, ci, co, 3, 1, 1, 3, 0.01, 512, 1, stream) == -2      # mode 3 without extra2: PP_ERR_BAD_ARG
    assert L.pp_conv_own_f16(vp(up.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(e1.data_ptr()), vp(y.data_ptr()),
                             n, h, w, ci, co, 3, 1, 1, 3, 0.01, 512, stream) == -2


@pytest.mark.parametrize("shape", [(2, 64, 128, 16, 16), (1, 128, 256, 8, 64), (3, 32, 64, 32, 32), (2, 96, 192, 6, 10),
                                   (1, 64, 128, 8, 128), (2, 128, 128, 32, 32)])
def test_collapsed_upsample_convolution_matches_the_convolution_of_the_upsampled_tensor(shape):
    """pp_conv_up2_collapsed_f16: conv3x3(upsample2(x)) evaluated as four 2x2 convolutions of x with per-phase tap sums
    (models/layers_transposed.py:270-275; 16 instead of 36 multiply-adds per input pixel).  Against the fp32 convolution of the
    materialised upsample with the ORIGINAL fp16 weights: the real-number result is the same, the fp16 rounding of the summed
    weights differs from the rounding of the individual ones -- 4e-3 of the output scale; image borders (zero padding of the
    upsampled tensor), one and two post adds, every tile width."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib, fused_model as fm
    L = _lib.load()
    n, ci, co, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(61)
    conv = torch.nn.Conv2d(ci, co, 3, 1, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (ci * 9) ** 0.5)
        conv.bias.copy_(torch.randn(co, generator=g))
    f = fm.FConv(conv, None, True).cuda().half()
    x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    e1 = torch.randn(n, co, 2 * h, 2 * w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    e2 = torch.randn(n, co, 2 * h, 2 * w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    up = F.interpolate(x.float(), scale_factor=2, mode="nearest")
    act = F.leaky_relu(F.conv2d(up, f.weight.float(), f.bias.float(), 1, 1), 0.01)
    w4 = f._collapsed_weights()
    assert w4.shape == (4, co, 2, 2, ci) and w4.dtype == torch.float16
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    for mode, ref in ((2, act + e1.float()), (3, act.half().float() + e1.float() + e2.float()), (0, act)):
        for bn in (256, 128, 64, 512):   # 512: the halo-tile kernel (maps it takes: width a multiple of 16, ...)
            if co % (bn if bn != 512 else 64) or (bn == 512 and (w % 16 or h % 8)):
                continue
            y = torch.full((n, co, 2 * h, 2 * w), float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            rc = L.pp_conv_up2_collapsed_f16(vp(x.data_ptr()), vp(w4.data_ptr()), vp(f.bias.data_ptr()), vp(e1.data_ptr()) if mode else None,
                                             vp(e2.data_ptr()) if mode == 3 else None, vp(y.data_ptr()), n, h, w, ci, co, mode, 0.01, bn, st)
            assert rc == 0, (shape, mode, bn, rc)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all(), (shape, mode, bn)
            err = (y.float() - ref).abs().max().item()
            assert err <= 4e-3 * max(1.0, ref.abs().max().item()), (shape, mode, bn, err)


def test_two_output_convolution_matches_convolution_plus_add():
    """pp_conv_own_ex_f16 mode 4 (every own kernel): y must equal the same kernel's mode-1 output bit for bit and y2 must be
    the binary16 sum y + extra2 (models/posenet.py:116-118: cache and x + cache); FConv.forward_dual returns the same pair as
    convolution followed by a tensor add."""
    import ctypes as C
    from posepaf import _lib, fused_model as fm
    L = _lib.load()
    vp = C.c_void_p
    g = torch.Generator(device="cpu").manual_seed(31)
    for (n, ci, co, h, w, k, pad), variants in [((2, 128, 256, 32, 32, 1, 0), (256, 128, 64, 0)), ((1, 64, 128, 16, 64, 3, 1), (128, 64, 512))]:
        x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, generator=g).cuda().half()
        e1 = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        e2 = torch.randn(n, co, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        stream = vp(torch.cuda.current_stream().cuda_stream)
        for bn in variants:
            y1 = torch.empty((n, co, h, w), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            assert L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(e1.data_ptr()), vp(y1.data_ptr()),
                                     n, h, w, ci, co, k, pad, 1, 1, 0.01, bn, stream) == 0
            y = torch.full_like(y1, float("nan"))
            y2 = torch.full_like(y1, float("nan"))
            assert L.pp_conv_own_ex_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(e1.data_ptr()), vp(e2.data_ptr()),
                                        vp(y.data_ptr()), vp(y2.data_ptr()), n, h, w, ci, co, k, pad, 1, 4, 0.01, bn, 0, stream) == 0
            torch.cuda.synchronize()
            assert torch.equal(y, y1), bn
            assert torch.equal(y2, y1 + e2), bn
        # missing second output / second operand: refused
        assert L.pp_conv_own_ex_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), vp(e1.data_ptr()), vp(e2.data_ptr()),
                                    vp(y.data_ptr()), None, n, h, w, ci, co, k, pad, 1, 4, 0.01, 0, 0, stream) == -2
    torch.manual_seed(6)
    f = fm.FConv(torch.nn.Conv2d(128, 256, 1, bias=True), None, False).cuda().half()
    x = torch.randn(2, 128, 32, 32, device="cuda").half().contiguous(memory_format=torch.channels_last)
    res = torch.randn(2, 256, 32, 32, device="cuda").half().contiguous(memory_format=torch.channels_last)
    other = torch.randn(2, 256, 32, 32, device="cuda").half().contiguous(memory_format=torch.channels_last)
    ref = f(x, res)
    key = ("dual", 2, 128, 32, 32, 256, 1, 0, 1, False, False, False)      # (..., SE gains folded, pooled third output)
    tol = 2e-3 * max(1.0, ref.float().abs().max().item())
    for choice in (0, 256, 128, fm.PW_VARIANT):
        fm._conv_choice[key] = choice
        y, y2 = f.forward_dual(x, res, other)
        assert (y.float() - ref.float()).abs().max().item() <= tol, choice
        assert torch.equal(y2, y + other), choice
    fm._conv_choice.pop(key)
    y, y2 = f.forward_dual(x, res, other)
    assert key in fm._conv_choice and torch.equal(y2, y + other)
    # the streaming 1x1 kernel's extras: the SE gains folded into the input read, and the 2x2 max-pool of y2 as a third output
    import torch.nn.functional as F
    gains = (0.3 + torch.rand(2, 128, device="cuda")).half()
    ref_s = f(fm.channel_scale(x, gains), res)
    keyp = ("dual", 2, 128, 32, 32, 256, 1, 0, 1, False, True, True)
    for choice in (0, fm.PW_VARIANT):
        fm._conv_choice[keyp] = choice
        y, y2, pooled = f.forward_dual(fm.Scaled(x, gains), res, other, want_pool=True)
        assert (y.float() - ref_s.float()).abs().max().item() <= tol, choice
        assert torch.equal(y2, y + other) and torch.equal(pooled, F.max_pool2d(y2, 2, 2)), choice


def test_fused_model_upsample_convolution_paths_agree():
    """FConv.forward_up2 (posepaf/fused_model.py): upsample -> convolution -> add(s) as separate launches and as ONE launch of
    the halo kernel give identical tensors, with one and with two added tensors; the hourglass uses whichever is faster."""
    from posepaf import fused_model as fm
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(128, 128, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(128)
    bn.running_mean.normal_(0, 0.1)
    bn.running_var.uniform_(0.5, 1.5)
    f = fm.FConv(conv, bn.eval(), True).cuda().half()
    low = torch.randn(2, 128, 32, 32, device="cuda").half().contiguous(memory_format=torch.channels_last)
    p1 = torch.randn(2, 128, 64, 64, device="cuda").half().contiguous(memory_format=torch.channels_last)
    p2 = torch.randn(2, 128, 64, 64, device="cuda").half().contiguous(memory_format=torch.channels_last)
    for post2 in (None, p2):
        sep = f(fm.upsample2(low), post=p1) if post2 is None else fm.add3(f(fm.upsample2(low)), p1, post2)
        key = ("up2", 2, 128, 32, 32, 128, post2 is not None, True)
        for choice in (0, 1):
            fm._conv_choice[key] = choice
            y = f.forward_up2(low, p1, post2)
            # the inner convolution of the separate path may run on another tile configuration than the halo kernel: same
            # operands, another summation order -> equal to within one binary16 rounding of the accumulator
            assert (y.float() - sep.float()).abs().max().item() <= 2e-3 * max(1.0, sep.float().abs().max().item()), (post2 is None, choice)
        fm._conv_choice.pop(key)
        y = f.forward_up2(low, p1, post2)     # timed choice
        assert key in fm._conv_choice and fm._conv_choice[key] in (0, 1)
        assert (y.float() - sep.float()).abs().max().item() <= 2e-3 * max(1.0, sep.float().abs().max().item())


def test_evaluate_two_ranks_on_one_gpu_gloo_rehearsal(tmp_path):
    """BASELINE configs[3]'s control flow on hardware: `evaluate.py --gpus 2` launches its own two ranks (one GPU shared, record
    exchange over gloo: POSEPAF_DIST_BACKEND=gloo), images sharded i mod 2, records gathered and re-interleaved into image
    order; the dump must equal the single-process run's."""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG
    outs = []
    for gpus in (1, 2):
        dump = tmp_path / f"res{gpus}.json"
        env = dict(os.environ, POSEPAF_DIST_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(PKG, "evaluate.py"), "--gpus", str(gpus), "--run_refactor", "--run_cpp",
                            "--synthetic", "5", "--sizes", "256x256", "--batch", "2", "--people", "3", "--dump_name", str(dump)],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        summary = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert summary["world"] == gpus and summary["images"] == 5 and summary["status_or"] == 0
        outs.append(json.load(open(dump)))
    # both runs execute the SAME kernel per layer shape (the first run saves its choice table, the second loads it and rank 0
    # broadcasts it) and every batch has the same geometry, so the dumps must be EQUAL, not merely close
    assert len(outs[0]) >= 10
    assert outs[0] == outs[1]


def test_evaluate_two_ranks_over_rccl_when_two_gpus_are_present(tmp_path):
    """BASELINE configs[3] on the real collective: `evaluate.py --gpus 2` over the nccl backend (= RCCL) when the box has two GPUs
    (an 8-GPU node runs this inside the GPU tier; a 1-GPU box skips it).  The dump must equal the 1-GPU run's."""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    outs = []
    for gpus in (1, 2):
        dump = tmp_path / f"res{gpus}.json"
        env = dict(os.environ)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "POSEPAF_DIST_BACKEND"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(PKG, "evaluate.py"), "--gpus", str(gpus), "--run_refactor", "--run_cpp",
                            "--synthetic", "12", "--sizes", "256x256", "--batch", "2", "--people", "3", "--dump_name", str(dump)],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        summary = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert summary["world"] == gpus and summary["images"] == 12 and summary["status_or"] == 0
        outs.append(json.load(open(dump)))
    assert outs[0] == outs[1]


def test_evaluate_original_path_with_a_scale_search(tmp_path):
    """evaluate.py WITHOUT --run_refactor (reference evaluate.py:81-89: predict + find_peaks + find_connections + find_humans) with
    two scales: pinned double-buffered staging, scene banks per scale, every scale accumulated by one launch; the injected people
    are recovered at image resolution (records carry PP_ST_FLOAT_COORDS = 32 and nothing else)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG
    dump = tmp_path / "res.json"
    r = subprocess.run([sys.executable, os.path.join(PKG, "evaluate.py"), "--synthetic", "5", "--sizes", "192x256", "--batch", "2",
                        "--scales", "0.5", "1.0", "--people", "2", "3", "--dump_name", str(dump)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    summary = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert summary["images"] == 5 and summary["status_or"] == 32 and summary["rules"] == "original"
    res = json.load(open(dump))
    assert sorted({r_["image_id"] for r_ in res}) == list(range(5)) and len(res) >= 8
    assert summary["synthetic_oks"]["AP"] > 0.5, summary


def test_evaluate_reads_a_coco_annotation_file_and_png_images(tmp_path):
    """The reference's data path (evaluate.py:72, :237-279): annotation JSON + image files in, COCO-format results and the
    keypoint AP against the ANNOTATION's ground truth out.  Offline there are no weights, so --inject_gt adds GT-style maps
    rendered from the annotation's own keypoints to the (live) network output; the people found must match the annotation."""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG
    from test_evaluate_sources_cpu import write_coco_fixture
    ann, img_dir = write_coco_fixture(str(tmp_path))
    dump = tmp_path / "res.json"
    r = subprocess.run([sys.executable, os.path.join(PKG, "evaluate.py"), "--run_refactor", "--run_cpp", "--ann_file", ann,
                        "--img_dir", img_dir, "--inject_gt", "--batch", "2", "--dump_name", str(dump)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    summary = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert summary["images"] == 3 and summary["status_or"] == 0        # the image without a person annotation is not evaluated
    res = json.load(open(dump))
    assert {r_["image_id"] for r_ in res} <= {1000, 1007, 1014} and len(res) >= 4
    assert all(len(r_["keypoints"]) == 51 and r_["category_id"] == 1 for r_ in res)
    assert summary["keypoint_ap"]["AP"] > 0.5 and summary["keypoint_ap"]["n_gt"] >= 4, summary


def test_ragged_preprocess_matches_the_per_image_padding():
    """pp_preprocess_u8_ragged: images of different sizes in one padded bucket == utils/util.py:44-65 per image (pad 128 on the
    bottom / right, / 255, mirror of the PADDED image); bytes outside an image's own area must not matter."""
    import ctypes as C
    from posepaf import _lib
    from posepaf.pipeline import preprocess_batch
    g = torch.Generator(device="cpu").manual_seed(11)
    hp, wp, sizes = 128, 192, [(128, 192), (100, 150), (65, 129), (1, 1)]
    slot = torch.randint(0, 256, (len(sizes), hp, wp, 3), dtype=torch.uint8, generator=g)       # garbage outside the images
    dims = torch.tensor([[h for h, _ in sizes], [w for _, w in sizes]], dtype=torch.int32).cuda()
    for dtype, code in ((torch.float16, _lib.PP_F16), (torch.float32, _lib.PP_F32)):
        out = torch.empty((2 * len(sizes), hp, wp, 3), dtype=dtype, device="cuda")
        _lib.check(_lib.load().pp_preprocess_u8_ragged(C.c_void_p(slot.cuda().data_ptr()), C.c_void_p(dims.data_ptr()),
                                                       C.c_void_p(out.data_ptr()), code, len(sizes), hp, wp, 128, 1,
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        for b, (h, w) in enumerate(sizes):
            want = torch.full((hp, wp, 3), 128.0) / 255.0
            want[:h, :w] = slot[b, :h, :w].float() / 255.0
            assert torch.equal(out[2 * b].cpu(), want.to(dtype))
            assert torch.equal(out[2 * b + 1].cpu(), want.flip(1).to(dtype))
    # equal sizes: the ragged kernel == the equal-size kernel
    imgs = slot[:, :128, :192].contiguous().cuda()
    d2 = torch.tensor([[128] * 4, [192] * 4], dtype=torch.int32).cuda()
    out = torch.empty((8, 128, 192, 3), dtype=torch.float16, device="cuda")
    _lib.check(_lib.load().pp_preprocess_u8_ragged(C.c_void_p(imgs.data_ptr()), C.c_void_p(d2.data_ptr()), C.c_void_p(out.data_ptr()),
                                                   _lib.PP_F16, 4, 128, 192, 128, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(out, preprocess_batch(imgs, True, torch.float16))


def test_util_keypoint_heatmap_nms_routes_to_the_hip_kernel():
    """utils.util.keypoint_heatmap_nms on a device tensor == the torch expression of utils/util.py:177-185 (reflect pad + 3x3
    max pool, >= thre), computed through K_A's 3x3 mode."""
    import torch.nn.functional as F
    from utils import util
    g = torch.Generator(device="cpu").manual_seed(3)
    heat = (torch.rand(1, 18, 24, 40, generator=g) ** 30).cuda()     # ~70 pixels per channel above the threshold
    heat[0, :, 0, 0] = 0.9
    heat[0, :, 23, 39] = 0.8
    heat[0, 3, 10, 10] = heat[0, 3, 10, 11] = 0.95           # plateau: both kept (>= comparison)
    assert util._hip_keypoint_nms(heat, 0.1) is not None      # the HIP path takes this map (no silent torch fallback)
    got = util.keypoint_heatmap_nms(heat, kernel=3, thre=0.1)
    hmax = F.max_pool2d(F.pad(heat, (1, 1, 1, 1), mode="reflect"), 3, stride=1)
    want = heat * ((hmax == heat).float() * (heat >= 0.1).float())
    assert torch.equal(got, want) and (got != 0).sum() > 300


@pytest.mark.parametrize("flags", [["--run_refactor", "--run_cpp"], ["--run_refactor"], []])
def test_demo_image_script_draws_the_injected_people(tmp_path, flags):
    """improved-body-parts_amd/demo_image.py (reference demo_image.py:80-321, same flags): one image through the three branches
    (pafprocess rules, Python rules, original path), rendering by utils/draw.py.  Offline the people come from a synthetic
    scene added to the network output; the canvas must differ from the input exactly where skeletons were drawn."""
    import os
    import subprocess
    import sys
    from conftest import PKG
    img = np.random.default_rng(5).integers(0, 256, (256, 256, 3), dtype=np.uint8)
    src, dst = tmp_path / "in.npy", tmp_path / "out.npy"
    np.save(src, img)
    r = subprocess.run([sys.executable, os.path.join(PKG, "demo_image.py"), "--image", str(src), "--output", str(dst), "--synthetic", "3",
                        *flags], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    n_people = int(r.stdout.strip().splitlines()[-1].split(",")[1].split()[0])
    assert n_people >= 2, r.stdout
    out = np.load(dst)
    assert out.shape == img.shape and out.dtype == np.uint8
    changed = (out != img).any(axis=2)
    assert 500 < changed.sum() < 0.5 * changed.size          # skeletons drawn, most of the image untouched
    # ... and drawn WHERE the people are: every joint of the injected scene (a local maximum of its keypoint channel, at a quarter
    # of the image resolution) has drawn pixels next to it -- in the reference's joint / pair colours on the refactored branches
    # (utils/common.py:240-264: discs and lines in CocoColors), blended limb ellipses on the original one (demo_image.py:174-240)
    from scipy import ndimage
    from posepaf import synth
    from utils import draw
    scene = synth.make_net_output(3, 4242, h=64, w=64, noise=0.0, dtype=np.float32, flip=False)[0]
    palette = {tuple(c) for c in draw.CocoColors}
    joints = hits = 0
    for part in range(18):
        ch = scene[30 + part]
        for y, x in zip(*np.nonzero((ch == ndimage.maximum_filter(ch, 5)) & (ch > 0.5))):
            win = (slice(max(0, 4 * y - 4), 4 * y + 9), slice(max(0, 4 * x - 4), 4 * x + 9))
            joints += 1
            if flags:
                hits += any(tuple(px) in palette for px in out[win][changed[win]])
            else:
                hits += bool(changed[win].any())
    assert joints >= 3 * 10 and hits >= 0.9 * joints, (joints, hits)


def test_inference_speed_script_prints_the_reference_log_lines():
    """improved-body-parts_amd/test_inference_speed.py (reference test_inference_speed.py:91-120): the forward-only loop with one
    synchronize per batch and the reference's `Test: [i/n] Time .. Speed ..` line; fused model under graph replay and the plain
    nn.Module eagerly."""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG
    for extra in (["--batch", "4", "--size", "256", "256"], ["--batch", "2", "--size", "128", "192", "--plain", "--no_graph"]):
        r = subprocess.run([sys.executable, os.path.join(PKG, "test_inference_speed.py"), "--iters", "4", "--json", "--opt-level", "O1",
                            *extra], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = r.stdout.strip().splitlines()
        assert sum(ln.startswith("==================>Test: [") and "Speed" in ln for ln in lines) == 4
        out = json.loads(lines[-1])
        assert out["value"] > 0 and out["batch"] == int(extra[1])


@pytest.mark.parametrize("shape", [(3, 128, 128, 50), (2, 24, 40, 50), (1, 7, 9, 64), (4, 16, 16, 1)])
def test_pixel_major_prediction_to_channel_planes(shape):
    """pp_nhwc64_to_planes_f16 / fused_model.to_planes: the [:, :c] view of a 64-channel channels-last tensor as the contiguous
    (n, c, h, w) tensor K_A / K_B read -- bit-equal to torch's strided copy, ragged tile ends included."""
    from posepaf import fused_model as fm
    n, h, w, c = shape
    base = torch.randn(n, 64, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
    view = base[:, :c]
    got = fm.to_planes(view)
    assert got.is_contiguous() and got.shape == (n, c, h, w) and torch.equal(got, view.contiguous())
    other = torch.randn(n, 48, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)[:, :40]
    assert torch.equal(fm.to_planes(other), other.contiguous())       # not the 64-channel layout: torch's copy


def test_own_convolution_kernels_are_deterministic_under_load():
    """Race detector.  The hand-written kernels keep LDS-DMA in flight across barriers and count it by hand (vmcnt); a wrong count
    shows only when the memory system is loaded -- the four-tap halo instances read a halo that was still arriving at 256
    samples while every small-shape test passed.  Every kernel family at the bench geometry, 256 samples, launched repeatedly
    on the same operands: all results BIT-identical, and the first right against torch on a slice."""
    import ctypes as C
    import torch.nn.functional as F
    from posepaf import _lib, fused_model as fm
    L = _lib.load()
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(97)
    n, reps = 256, 6

    def repeat(launch, shape_out, check):
        first = None
        for r in range(reps):
            y = torch.full(shape_out, float("nan"), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
            assert launch(y) == 0
            torch.cuda.synchronize()
            if first is None:
                first = y
                assert torch.isfinite(y).all()
                check(y)
            else:
                assert torch.equal(y, first), ("repeat differs", r, float((y.float() - first.float()).abs().max()))

    # collapsed upsample convolution, four-tap halo instances (64-, 32- and 16-wide tiles) and implicit GEMM
    for ci, co, h, w in ((256, 256, 64, 64), (384, 384, 32, 32), (512, 512, 16, 16)):
        conv = torch.nn.Conv2d(ci, co, 3, 1, 1, bias=True)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (ci * 9) ** 0.5)
            conv.bias.copy_(torch.randn(co, generator=g))
        f = fm.FConv(conv, None, True).cuda().half()
        x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        e1 = torch.randn(n, co, 2 * h, 2 * w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        w4 = f._collapsed_weights()
        ref = F.leaky_relu(F.conv2d(F.interpolate(x[:2].float(), scale_factor=2, mode="nearest"), f.weight.float(), f.bias.float(), 1, 1),
                           0.01) + e1[:2].float()
        for bn in (512, 128):
            repeat(lambda y: L.pp_conv_up2_collapsed_f16(vp(x.data_ptr()), vp(w4.data_ptr()), vp(f.bias.data_ptr()), vp(e1.data_ptr()), None,
                                                         vp(y.data_ptr()), n, h, w, ci, co, 2, 0.01, bn, st),
                   (n, co, 2 * h, 2 * w), lambda y: (y[:2].float() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item() or
                   pytest.fail("collapsed convolution wrong"))
        del x, e1
    # nine-tap halo kernel (128- and 64-wide tiles), dilated instances, implicit GEMM, streaming 1x1
    for ci, co, h, w, k, d in ((128, 128, 128, 128, 3, 1), (192, 192, 64, 64, 3, 1), (128, 128, 128, 128, 3, 4), (256, 128, 64, 64, 1, 1)):
        x = torch.randn(n, ci, h, w, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).cuda().half().contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, generator=g).cuda().half()
        pad = d if k == 3 else 0
        ref = F.leaky_relu(F.conv2d(x[:2].float(), wt.float(), b.float(), 1, pad, d), 0.01)

        def check(y):
            assert (y[:2].float() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
        for bn in ((512, 128 if co % 128 == 0 else 64) if k == 3 else (128,)):
            repeat(lambda y: L.pp_conv_own_f16(vp(x.data_ptr()), vp(wt.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), n, h, w, ci, co,
                                               k, pad, d, 0, 0.01, bn, st), (n, co, h, w), check)
        if k == 1:
            repeat(lambda y: L.pp_pw_f16(vp(x.data_ptr()), None, vp(wt.data_ptr()), vp(b.data_ptr()), None, None, vp(y.data_ptr()), None,
                                         n * h * w, h * w, ci, co, co, 0, 0.01, st), (n, co, h, w), check)
        del x


@pytest.mark.parametrize("shape", [(5, 256, 16, 40), (3, 768, 48, 7), (2, 384, 24, 1)])
def test_se_excitation_kernel_matches_the_torch_modules(shape):
    """pp_se_gains_f16 (models/layers_transposed.py:289-310: Linear -> LeakyReLU -> Linear -> Sigmoid on the channel means) from
    partial channel sums and from a ready mean, against the fp16 torch modules on the same mean: the roundings sit at the same
    places, the fp32 sums run in another order -- one binary16 ulp of a value below 1."""
    import ctypes as C
    from posepaf import _lib
    L = _lib.load()
    n, c, hid, splits = shape
    g = torch.Generator(device="cpu").manual_seed(41)
    fc1, fc2 = torch.nn.Linear(c, hid), torch.nn.Linear(hid, c)
    with torch.no_grad():
        fc1.weight.copy_(torch.randn(fc1.weight.shape, generator=g) / c ** 0.5), fc1.bias.copy_(torch.randn(hid, generator=g))
        fc2.weight.copy_(torch.randn(fc2.weight.shape, generator=g)), fc2.bias.copy_(torch.randn(c, generator=g))
    fc1, fc2 = fc1.cuda().half(), fc2.cuda().half()
    hw = 4096
    ws = (torch.randn(n, splits, c, generator=g) * 40 + 10).cuda()
    mean = (ws.sum(dim=1) / hw).half()
    with torch.no_grad():
        want = torch.sigmoid(fc2(torch.nn.functional.leaky_relu(fc1(mean), 0.01))).float()
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    args = (vp(fc1.weight.data_ptr()), vp(fc1.bias.data_ptr()), vp(fc2.weight.data_ptr()), vp(fc2.bias.data_ptr()))
    a = torch.full((n, c), float("nan"), dtype=torch.float16, device="cuda")
    b = torch.full((n, c), float("nan"), dtype=torch.float16, device="cuda")
    assert L.pp_se_gains_f16(vp(ws.data_ptr()), None, *args, vp(a.data_ptr()), n, hw, c, hid, splits, 0.01, st) == 0
    assert L.pp_se_gains_f16(None, vp(mean.data_ptr()), *args, vp(b.data_ptr()), n, 0, c, hid, 0, 0.01, st) == 0
    torch.cuda.synchronize()
    assert (a.float() - want).abs().max().item() <= 1.5e-3 and (b.float() - want).abs().max().item() <= 1.5e-3
    assert L.pp_se_gains_f16(vp(ws.data_ptr()), vp(mean.data_ptr()), *args, vp(a.data_ptr()), n, hw, c, hid, splits, 0.01, st) == -2
