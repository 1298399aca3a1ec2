// posepaf_epilogue.hip -- fused convolution epilogue for the IMHN forward (channels-last fp16):
//     y = act(y + bias[c] (+ residual))       act = LeakyReLU(slope) or identity
// One pass over the activation instead of the three PyTorch-ROCm issues per convolution (MIOpen bias OpTensor,
// a strided add, leaky_relu): the convolution itself stays on MIOpen (BASELINE.json north_star), this removes
// ~40 % of the forward's kernel time that was pure elementwise traffic (profiles/r01_e2e_b8_kernel_stats.csv).
// 16-byte vectors (8 halves); NHWC storage so the channel is the fastest index and bias vectors are contiguous.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/posepaf.h"

namespace {

template <bool HAS_RES, bool HAS_ACT>
__global__ __launch_bounds__(256) void k_bias_act(uint4 *__restrict__ y, const uint4 *__restrict__ bias,
                                                  const uint4 *__restrict__ res, long nvec, int cvec, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint4 a = y[v];
        const uint4 b = bias[v % cvec];
        uint4 r;
        if (HAS_RES) r = res[v];
        __half2 *ah = reinterpret_cast<__half2 *>(&a);
        const __half2 *bh = reinterpret_cast<const __half2 *>(&b);
        const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float2 f = __half22float2(ah[k]);
            const float2 fb = __half22float2(bh[k]);
            f.x += fb.x;
            f.y += fb.y;
            if (HAS_RES) {
                const float2 fr = __half22float2(rh[k]);
                f.x += fr.x;
                f.y += fr.y;
            }
            if (HAS_ACT) {
                f.x = f.x > 0.f ? f.x : f.x * slope;
                f.y = f.y > 0.f ? f.y : f.y * slope;
            }
            ah[k] = __float22half2_rn(f);
        }
        y[v] = a;
    }
}

}  // namespace

extern "C" int pp_bias_act_f16(void *y, const void *bias, const void *residual, long n_elems, int channels, float slope,
                               int has_act, void *stream) {
    if (!y || !bias || n_elems <= 0 || channels <= 0 || (channels & 7) || (n_elems % channels) ||
        (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) ||
        (reinterpret_cast<uintptr_t>(residual) & 15))
        return PP_ERR_BAD_ARG;
    const long nvec = n_elems / 8;
    const int cvec = channels / 8;
    long blocks = (nvec + 255) / 256;
    if (blocks > 2048 * 4) blocks = 2048 * 4;  // grid-stride beyond 8 blocks per CU x 4
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint4 *yy = static_cast<uint4 *>(y);
    const uint4 *bb = static_cast<const uint4 *>(bias), *rr = static_cast<const uint4 *>(residual);
    if (residual) {
        if (has_act) hipLaunchKernelGGL((k_bias_act<true, true>), dim3(blocks), dim3(256), 0, st, yy, bb, rr, nvec, cvec, slope);
        else hipLaunchKernelGGL((k_bias_act<true, false>), dim3(blocks), dim3(256), 0, st, yy, bb, rr, nvec, cvec, slope);
    } else {
        if (has_act) hipLaunchKernelGGL((k_bias_act<false, true>), dim3(blocks), dim3(256), 0, st, yy, bb, rr, nvec, cvec, slope);
        else hipLaunchKernelGGL((k_bias_act<false, false>), dim3(blocks), dim3(256), 0, st, yy, bb, rr, nvec, cvec, slope);
    }
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}
