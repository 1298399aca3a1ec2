// A1 forward, fused convolution:  y = leaky(conv(x, w) + bias (+ residual)) (+ post)  in ONE kernel.
//
// The implicit-GEMM main loop is composable_kernel's DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle (ROCm's CK headers,
// the same XDL/MFMA kernels MIOpen dispatches to for these shapes -- see profiles/), instantiated here with this
// project's own epilogue functor so that the folded-BN bias, the LeakyReLU, the residual add and the hourglass "up1 +"
// add are applied to the fp32 accumulators in registers instead of in a second pass over the activation
// (csrc/posepaf_epilogue.hip, 17 % of a step before).  One translation unit per tile configuration (-DPP_CONV_CFG=n):
// the tile shapes are the f16 entries of CK's own instance list; which one runs for a given layer is measured at first use
// (posepaf/fused_model.py).  Reference: models/layers_transposed.py Conv / Residual / Hourglass (Conv2d + BN + LeakyReLU).
#include <array>
#include <hip/hip_runtime.h>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/convolution_forward_specialization.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle_v3.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"

#include "posepaf_conv.h"

#ifndef PP_CONV_CFG
#error "compile with -DPP_CONV_CFG=<0..9>"
#endif

namespace {
using F16 = ck::half_t;
using F32 = float;
template <ck::index_t... Is>
using S = ck::Sequence<Is...>;
using namespace ck::tensor_layout::convolution;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
using ck::tensor_operation::device::ConvolutionForwardSpecialization;
using ck::tensor_operation::device::GemmSpecialization;

// epilogue on the fp32 accumulator c; d0 = bias[k]; d1 = residual (pre != 0) or post-add (pre == 0)
struct BiasLeaky {
    float slope;
    template <typename E, typename C, typename D0>
    __host__ __device__ void operator()(E &e, const C &c, const D0 &d0) const {
        float v = c + ck::type_convert<float>(d0);
        v = v > 0.f ? v : v * slope;
        e = ck::type_convert<E>(v);
    }
};
struct BiasAddLeaky {
    float slope;
    int pre;
    template <typename E, typename C, typename D0, typename D1>
    __host__ __device__ void operator()(E &e, const C &c, const D0 &d0, const D1 &d1) const {
        float v = c + ck::type_convert<float>(d0);
        const float x = ck::type_convert<float>(d1);
        if (pre) v += x;
        v = v > 0.f ? v : v * slope;
        if (!pre) v += x;
        e = ck::type_convert<E>(v);
    }
};

// tile configurations: BlockSize, MPerBlock, NPerBlock, MXdlPerWave, NXdlPerWave, K0 x threads cluster, C-shuffle cluster
#if PP_CONV_CFG == 0
#define PP_TILE 256, 128, 256, 32, 8, 8, 32, 32, 2, 4
#define PP_CLUSTER S<4, 64, 1>
#define PP_CSHUF S<1, 32, 1, 8>
#elif PP_CONV_CFG == 1
#define PP_TILE 256, 256, 128, 32, 8, 8, 32, 32, 4, 2
#define PP_CLUSTER S<4, 64, 1>
#define PP_CSHUF S<1, 32, 1, 8>
#elif PP_CONV_CFG == 2
#define PP_TILE 256, 128, 128, 32, 8, 8, 32, 32, 2, 2
#define PP_CLUSTER S<4, 64, 1>
#define PP_CSHUF S<1, 32, 1, 8>
#elif PP_CONV_CFG == 3
#define PP_TILE 128, 128, 64, 32, 8, 8, 32, 32, 2, 2
#define PP_CLUSTER S<4, 32, 1>
#define PP_CSHUF S<1, 32, 1, 4>
#elif PP_CONV_CFG == 4
#define PP_TILE 256, 128, 64, 32, 8, 8, 32, 32, 2, 1
#define PP_CLUSTER S<4, 64, 1>
#define PP_CSHUF S<1, 32, 1, 8>
#elif PP_CONV_CFG == 5
#define PP_TILE 256, 64, 128, 32, 8, 8, 32, 32, 1, 2
#define PP_CLUSTER S<4, 64, 1>
#define PP_CSHUF S<1, 32, 1, 8>
#elif PP_CONV_CFG == 6
// 6..9: the newer "V3" device op (software-pipelined block GEMM)
#define PP_V3
#define PP_TILE 256, 128, 128, 64, 8, 8, 32, 32, 2, 2
#define PP_CLUSTER S<8, 32, 1>
#define PP_LDSPAD 0
#define PP_CSHUF S<1, 32, 1, 8>
#define PP_PIPE ck::BlockGemmPipelineScheduler::Intrawave, ck::BlockGemmPipelineVersion::v4
#elif PP_CONV_CFG == 7
#define PP_V3
#define PP_TILE 256, 128, 128, 64, 8, 8, 32, 32, 2, 2
#define PP_CLUSTER S<8, 32, 1>
#define PP_LDSPAD 0
#define PP_CSHUF S<1, 32, 1, 8>
#define PP_PIPE ck::BlockGemmPipelineScheduler::Intrawave, ck::BlockGemmPipelineVersion::v3
#elif PP_CONV_CFG == 8
#define PP_V3
#define PP_TILE 256, 256, 256, 32, 8, 8, 32, 32, 4, 4
#define PP_CLUSTER S<4, 64, 1>
#define PP_LDSPAD 0
#define PP_CSHUF S<1, 32, 1, 8>
#define PP_PIPE ck::BlockGemmPipelineScheduler::Intrawave, ck::BlockGemmPipelineVersion::v3
#elif PP_CONV_CFG == 9
#define PP_V3
#define PP_TILE 256, 256, 256, 64, 8, 8, 32, 32, 4, 4
#define PP_CLUSTER S<8, 32, 1>
#define PP_LDSPAD 0
#define PP_CSHUF S<1, 32, 1, 8>
#define PP_PIPE ck::BlockGemmPipelineScheduler::Intrawave, ck::BlockGemmPipelineVersion::v3
#endif

#ifdef PP_V3
template <typename DsLayout, typename DsTypes, typename Epilogue>
using Conv = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle_V3<
    2, NHWGC, GKYXC, DsLayout, NHWGK, F16, F16, F32, F16, DsTypes, F16, PassThrough, PassThrough, Epilogue,
    ConvolutionForwardSpecialization::Default, GemmSpecialization::MNKPadding, PP_TILE, PP_CLUSTER, S<1, 0, 2>, S<1, 0, 2>, 2, 8,
    8, PP_LDSPAD, PP_CLUSTER, S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, PP_LDSPAD, 1, 1, PP_CSHUF, 8, PP_PIPE>;
#else
template <typename DsLayout, typename DsTypes, typename Epilogue>
using Conv = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<
    2, NHWGC, GKYXC, DsLayout, NHWGK, F16, F16, F32, F16, DsTypes, F16, PassThrough, PassThrough, Epilogue,
    ConvolutionForwardSpecialization::Default, GemmSpecialization::MNKPadding, 1, PP_TILE, PP_CLUSTER, S<1, 0, 2>, S<1, 0, 2>,
    2, 8, 8, 1, PP_CLUSTER, S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, 1, 1, 1, PP_CSHUF, 8>;
#endif

using ConvBias = Conv<ck::Tuple<G_K>, ck::Tuple<F16>, BiasLeaky>;
using ConvBiasAdd = Conv<ck::Tuple<G_K, NHWGK>, ck::Tuple<F16, F16>, BiasAddLeaky>;

// One launch over `n` images starting at image `n0` of the batch.
int run_part(const PPConvArgs &a, int n0, int n, bool launch) {
    using idx = ck::index_t;
    const idx G = 1, N = n, H = a.H, W = a.W, C = a.C, K = a.K, R = a.R, Sx = a.S;
    const idx Ho = H + 2 * a.pad - a.dil * (R - 1), Wo = W + 2 * a.pad - a.dil * (Sx - 1);
    const idx LX = a.ldx, LY = a.ldy;   // pixel strides of x / y: C / K when packed, larger for a channel slice of a wider tensor
    const size_t in_off = (size_t)n0 * H * W * LX * sizeof(F16), out_off = (size_t)n0 * Ho * Wo * K * sizeof(F16);
    const void *x = static_cast<const char *>(a.x) + in_off;
    const void *extra = a.extra ? static_cast<const char *>(a.extra) + out_off : nullptr;
    void *y = static_cast<char *>(a.y) + (size_t)n0 * Ho * Wo * LY * sizeof(F16);
    // lengths in (G, N, C|K, spatial...) order; strides of the NHWGC / GKYXC / NHWGK tensors (G = 1); x_str: the packed extra
    const std::array<idx, 5> a_len{G, N, C, H, W}, a_str{C, H * W * LX, 1, W * LX, LX};
    const std::array<idx, 5> b_len{G, K, C, R, Sx}, b_str{K * R * Sx * C, R * Sx * C, 1, Sx * C, C};
    const std::array<idx, 5> e_len{G, N, K, Ho, Wo}, e_str{K, Ho * Wo * LY, 1, Wo * LY, LY};
    const std::array<idx, 5> x_str{K, Ho * Wo * K, 1, Wo * K, K};
    const std::array<idx, 5> bias_str{K, 0, 1, 0, 0};
    const std::array<idx, 2> strides{1, 1}, dil{a.dil, a.dil}, pads{a.pad, a.pad};
    const StreamConfig cfg{static_cast<hipStream_t>(a.stream), false};
    if (a.extra_mode == 0) {
        ConvBias op;
        auto arg = op.MakeArgument(x, a.w, std::array<const void *, 1>{a.bias}, y, a_len, a_str, b_len, b_str,
                                   std::array<std::array<idx, 5>, 1>{e_len}, std::array<std::array<idx, 5>, 1>{bias_str}, e_len,
                                   e_str, strides, dil, pads, pads, PassThrough{}, PassThrough{}, BiasLeaky{a.slope});
        if (!op.IsSupportedArgument(arg)) return -1;
        if (launch) op.MakeInvoker().Run(arg, cfg);
    } else {
        ConvBiasAdd op;
        auto arg = op.MakeArgument(x, a.w, std::array<const void *, 2>{a.bias, extra}, y, a_len, a_str, b_len, b_str,
                                   std::array<std::array<idx, 5>, 2>{e_len, e_len},
                                   std::array<std::array<idx, 5>, 2>{bias_str, x_str}, e_len, e_str, strides, dil, pads, pads,
                                   PassThrough{}, PassThrough{}, BiasAddLeaky{a.slope, a.extra_mode == 1});
        if (!op.IsSupportedArgument(arg)) return -1;
        if (launch) op.MakeInvoker().Run(arg, cfg);
    }
    return 0;
}

// The device templates bound some 32-bit GEMM extents (e.g. M * K * 2 bytes <= 2 GB in the pipelined variants); a batch that
// exceeds them is run as 2, 4 or 8 equal sub-batches (images are independent rows of the implicit GEMM).
int run(const PPConvArgs &a) {
    for (int parts = 1; parts <= 8; parts *= 2) {
        if (a.N % parts) break;
        const int n = a.N / parts;
        if (run_part(a, 0, n, false) != 0) continue;
        for (int p = 0; p < parts; p++)
            if (run_part(a, p * n, n, true) != 0) return -1;
        return 0;
    }
    return -1;
}
}  // namespace

#define PP_CAT2(a, b) a##b
#define PP_CAT(a, b) PP_CAT2(a, b)
int PP_CAT(pp_conv_run_cfg, PP_CONV_CFG)(const PPConvArgs &a) { return run(a); }
