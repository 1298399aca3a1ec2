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
