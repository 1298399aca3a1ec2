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

// y = act(y + bias[c] (+ res)) (+ post)      res is added BEFORE the activation (residual blocks), post AFTER it
// (hourglass: up1 + conv(upsample(..)))
template <bool HAS_RES, bool HAS_ACT, bool HAS_POST>
__global__ __launch_bounds__(256) void k_bias_act(uint4 *__restrict__ y, const uint4 *__restrict__ bias,
                                                  const uint4 *__restrict__ res, const uint4 *__restrict__ post, long nvec,
                                                  int cvec, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint4 a = y[v];
        const uint4 b = bias[v % cvec];
        uint4 r, p;
        if (HAS_RES) r = res[v];
        if (HAS_POST) p = post[v];
        __half2 *ah = reinterpret_cast<__half2 *>(&a);
        const __half2 *bh = reinterpret_cast<const __half2 *>(&b);
        const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
        const __half2 *ph = reinterpret_cast<const __half2 *>(&p);
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
            if (HAS_POST) {
                const float2 fp = __half22float2(ph[k]);
                f.x += fp.x;
                f.y += fp.y;
            }
            ah[k] = __float22half2_rn(f);
        }
        y[v] = a;
    }
}

// 2x2/2 max pool and x2 nearest upsample on NHWC fp16, 8 channels (16 bytes) per lane, coalesced along C.
__global__ __launch_bounds__(256) void k_maxpool2(const uint4 *__restrict__ x, uint4 *__restrict__ y, long nvec_out, int Ho,
                                                  int Wo, int cvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    const int W = Wo * 2;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec_out; v += stride) {
        const int c = (int)(v % cvec);
        long t = v / cvec;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        const long base = ((n * (Ho * 2) + ho * 2) * W + wo * 2) * cvec + c;
        const uint4 a = x[base], b = x[base + cvec], cc = x[base + (long)W * cvec], d = x[base + (long)W * cvec + cvec];
        uint4 o;
        const __half2 *ah = reinterpret_cast<const __half2 *>(&a), *bh = reinterpret_cast<const __half2 *>(&b);
        const __half2 *ch = reinterpret_cast<const __half2 *>(&cc), *dh = reinterpret_cast<const __half2 *>(&d);
        __half2 *oh = reinterpret_cast<__half2 *>(&o);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float2 fa = __half22float2(ah[k]), fb = __half22float2(bh[k]);
            const float2 fc = __half22float2(ch[k]), fd = __half22float2(dh[k]);
            oh[k] = __float22half2_rn(make_float2(fmaxf(fmaxf(fa.x, fb.x), fmaxf(fc.x, fd.x)),
                                                  fmaxf(fmaxf(fa.y, fb.y), fmaxf(fc.y, fd.y))));
        }
        y[v] = o;
    }
}

__global__ __launch_bounds__(256) void k_upsample2(const uint4 *__restrict__ x, uint4 *__restrict__ y, long nvec_in, int Hi,
                                                   int Wi, int cvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    const int W = Wi * 2;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec_in; v += stride) {
        const int c = (int)(v % cvec);
        long t = v / cvec;
        const int wi = (int)(t % Wi);
        t /= Wi;
        const int hi = (int)(t % Hi);
        const long n = t / Hi;
        const uint4 a = x[v];
        const long base = ((n * (Hi * 2) + hi * 2) * W + wi * 2) * cvec + c;
        y[base] = a;
        y[base + cvec] = a;
        y[base + (long)W * cvec] = a;
        y[base + (long)W * cvec + cvec] = a;
    }
}

inline long grid_for(long nvec) {
    long blocks = (nvec + 255) / 256;
    return blocks > 8192 ? 8192 : blocks;
}

}  // namespace

extern "C" int pp_bias_act_f16(void *y, const void *bias, const void *residual, const void *post, long n_elems,
                               int channels, float slope, int has_act, void *stream) {
    if (!y || !bias || n_elems <= 0 || channels <= 0 || (channels & 7) || (n_elems % channels) ||
        (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) ||
        (reinterpret_cast<uintptr_t>(residual) & 15) || (reinterpret_cast<uintptr_t>(post) & 15))
        return PP_ERR_BAD_ARG;
    const long nvec = n_elems / 8;
    const int cvec = channels / 8;
    const dim3 grid(grid_for(nvec)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint4 *yy = static_cast<uint4 *>(y);
    const uint4 *bb = static_cast<const uint4 *>(bias), *rr = static_cast<const uint4 *>(residual);
    const uint4 *pp = static_cast<const uint4 *>(post);
#define PP_LAUNCH(R, A, P) hipLaunchKernelGGL((k_bias_act<R, A, P>), grid, block, 0, st, yy, bb, rr, pp, nvec, cvec, slope)
    const int sel = (residual ? 4 : 0) | (has_act ? 2 : 0) | (post ? 1 : 0);
    switch (sel) {
        case 0: PP_LAUNCH(false, false, false); break;
        case 1: PP_LAUNCH(false, false, true); break;
        case 2: PP_LAUNCH(false, true, false); break;
        case 3: PP_LAUNCH(false, true, true); break;
        case 4: PP_LAUNCH(true, false, false); break;
        case 5: PP_LAUNCH(true, false, true); break;
        case 6: PP_LAUNCH(true, true, false); break;
        default: PP_LAUNCH(true, true, true); break;
    }
#undef PP_LAUNCH
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

extern "C" int pp_maxpool2_f16(const void *x, void *y, long n, int h_out, int w_out, int channels, void *stream) {
    if (!x || !y || n <= 0 || h_out <= 0 || w_out <= 0 || channels <= 0 || (channels & 7) ||
        (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
        return PP_ERR_BAD_ARG;
    const long nvec = n * h_out * w_out * (channels / 8);
    hipLaunchKernelGGL(k_maxpool2, dim3(grid_for(nvec)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(x), static_cast<uint4 *>(y), nvec, h_out, w_out, channels / 8);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

extern "C" int pp_upsample2_f16(const void *x, void *y, long n, int h_in, int w_in, int channels, void *stream) {
    if (!x || !y || n <= 0 || h_in <= 0 || w_in <= 0 || channels <= 0 || (channels & 7) ||
        (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
        return PP_ERR_BAD_ARG;
    const long nvec = n * h_in * w_in * (channels / 8);
    hipLaunchKernelGGL(k_upsample2, dim3(grid_for(nvec)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(x), static_cast<uint4 *>(y), nvec, h_in, w_in, channels / 8);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ three-way add
// y = a + b (+ c), fp32 sum rounded once: the hourglass's `up1 + up2` and the next stage's `+ cache` in one pass over the
// activation (posepaf/fused_model.py FHourglass; reference: models/layers_transposed.py Hourglass.forward, models/posenet.py)
namespace {
template <bool HAS_C>
__global__ __launch_bounds__(256) void k_add3(const uint4 *__restrict__ a, const uint4 *__restrict__ b, const uint4 *__restrict__ c,
                                              uint4 *__restrict__ y, long nvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const uint4 qa = a[v], qb = b[v];
        uint4 qc, r;
        if (HAS_C) qc = c[v];
        const __half2 *ha = reinterpret_cast<const __half2 *>(&qa), *hb = reinterpret_cast<const __half2 *>(&qb);
        const __half2 *hc = reinterpret_cast<const __half2 *>(&qc);
        __half2 *hr = reinterpret_cast<__half2 *>(&r);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float2 f = __half22float2(ha[k]);
            const float2 g = __half22float2(hb[k]);
            f.x += g.x;
            f.y += g.y;
            if (HAS_C) {
                const float2 h = __half22float2(hc[k]);
                f.x += h.x;
                f.y += h.y;
            }
            hr[k] = __float22half2_rn(f);
        }
        y[v] = r;
    }
}
}  // namespace

extern "C" int pp_add3_f16(const void *a, const void *b, const void *c, void *y, long n_elems, void *stream) {
    if (!a || !b || !y || n_elems <= 0 || (n_elems & 7) ||
        ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
          reinterpret_cast<uintptr_t>(y)) & 15))
        return PP_ERR_BAD_ARG;
    const long nvec = n_elems / 8;
    const dim3 grid(grid_for(nvec)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c)
        hipLaunchKernelGGL(k_add3<true>, grid, block, 0, st, static_cast<const uint4 *>(a), static_cast<const uint4 *>(b),
                           static_cast<const uint4 *>(c), static_cast<uint4 *>(y), nvec);
    else
        hipLaunchKernelGGL(k_add3<false>, grid, block, 0, st, static_cast<const uint4 *>(a), static_cast<const uint4 *>(b),
                           static_cast<const uint4 *>(nullptr), static_cast<uint4 *>(y), nvec);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ SE squeeze
// Channel means of an NHWC fp16 activation (models/layers_transposed.py SELayer: AdaptiveAvgPool2d(1)): two passes.
// (1) grid (splits, n): every workgroup sums its slice of the pixels, 8 channels (one 16-byte vector) per thread and
//     256 / (c/8) pixels in flight per step, fp32 accumulation, LDS reduction over the pixel lanes -> partial[n][split][c];
// (2) one thread per (n, c): sum of the splits / (h*w) -> fp16.
namespace {
__global__ __launch_bounds__(256) void k_channel_sum(const uint4 *__restrict__ x, float *__restrict__ partial, long hw, int cvec,
                                                     int splits) {
    __shared__ float s_acc[256 * 8];
    const int n = blockIdx.y, split = blockIdx.x;
    const int plan = 256 / cvec;                 // pixel lanes
    const int v = threadIdx.x % cvec, pl = threadIdx.x / cvec;
    const long per = (hw + splits - 1) / splits, p0 = (long)split * per, p1 = p0 + per < hw ? p0 + per : hw;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pl < plan) {
        const uint4 *base = x + (size_t)n * hw * cvec + v;
#pragma unroll 4
        for (long p = p0 + pl; p < p1; p += plan) {
            const uint4 q = base[(size_t)p * cvec];
            const __half2 *h = reinterpret_cast<const __half2 *>(&q);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float2 f = __half22float2(h[k]);
                acc[2 * k] += f.x;
                acc[2 * k + 1] += f.y;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) s_acc[threadIdx.x * 8 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < cvec) {  // pixel lane 0 of every channel vector adds the other lanes
        float *out = partial + ((size_t)n * splits + split) * cvec * 8 + (size_t)threadIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float t = 0.f;
            for (int l = 0; l < plan; l++) t += s_acc[(l * cvec + threadIdx.x) * 8 + k];
            out[k] = t;
        }
    }
}
__global__ __launch_bounds__(256) void k_channel_mean_finish(const float *__restrict__ partial, __half *__restrict__ out, int n_images,
                                                             int c, int splits, float inv_hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images * c) return;
    const int n = i / c, ch = i - n * c;
    float t = 0.f;
    for (int s = 0; s < splits; s++) t += partial[((size_t)n * splits + s) * c + ch];
    out[i] = __float2half_rn(t * inv_hw);
}
}  // namespace

// SE excitation: y[n][p][c] = x[n][p][c] * scale[n][c], 16-byte vectors, product rounded once to fp16 like torch's fp16 multiply
namespace {
__global__ __launch_bounds__(256) void k_channel_scale(const uint4 *__restrict__ x, const uint4 *__restrict__ scale, uint4 *__restrict__ y,
                                                       long hw, int cvec) {
    // grid (pixel blocks, n): a thread keeps ONE channel vector (its scale stays in registers) and walks over pixels
    const int n = blockIdx.y, plan = 256 / cvec, v = threadIdx.x % cvec, pl = threadIdx.x / cvec;
    if (pl >= plan) return;
    const uint4 b = scale[(size_t)n * cvec + v];
    const __half2 *bh = reinterpret_cast<const __half2 *>(&b);
    float2 fb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) fb[k] = __half22float2(bh[k]);
    const size_t base = (size_t)n * hw * cvec + v;
#pragma unroll 4
    for (long p = (long)blockIdx.x * plan + pl; p < hw; p += (long)gridDim.x * plan) {
        const uint4 a = x[base + (size_t)p * cvec];
        uint4 r;
        const __half2 *ah = reinterpret_cast<const __half2 *>(&a);
        __half2 *rh = reinterpret_cast<__half2 *>(&r);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float2 fa = __half22float2(ah[k]);
            rh[k] = __float22half2_rn(make_float2(fa.x * fb[k].x, fa.y * fb[k].y));
        }
        y[base + (size_t)p * cvec] = r;
    }
}
}  // namespace

extern "C" int pp_channel_scale_f16(const void *x, const void *scale, void *y, int n, long hw, int channels, void *stream) {
    if (!x || !scale || !y || n <= 0 || hw <= 0 || channels <= 0 || (channels & 7) ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(y)) & 15))
        return PP_ERR_BAD_ARG;
    const int cvec = channels / 8;
    if (cvec > 256 || n > 65535) return PP_ERR_BAD_ARG;
    const long per_block = 256 / cvec;  // pixels per workgroup step
    long bx = (hw + per_block * 8 - 1) / (per_block * 8);  // about 8 steps per thread
    bx = bx < 1 ? 1 : (bx > 4096 ? 4096 : bx);
    hipLaunchKernelGGL(k_channel_scale, dim3((unsigned)bx, n), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(x), static_cast<const uint4 *>(scale), static_cast<uint4 *>(y), hw, cvec);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

extern "C" int pp_channel_mean_f16(const void *x, void *partial_ws, void *out, int n, long hw, int channels, int splits,
                                   void *stream) {
    if (!x || !partial_ws || !out || n <= 0 || hw <= 0 || channels <= 0 || (channels & 7) || channels > 2048 || splits <= 0 ||
        splits > 65535 || (reinterpret_cast<uintptr_t>(x) & 15))
        return PP_ERR_BAD_ARG;
    const int cvec = channels / 8;
    if (cvec > 256) return PP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_channel_sum, dim3(splits, n), dim3(256), 0, st, static_cast<const uint4 *>(x),
                       static_cast<float *>(partial_ws), hw, cvec, splits);
    const int total = n * channels;
    hipLaunchKernelGGL(k_channel_mean_finish, dim3((total + 255) / 256), dim3(256), 0, st, static_cast<const float *>(partial_ws),
                       static_cast<__half *>(out), n, channels, splits, 1.0f / (float)hw);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// the second half of pp_channel_mean_f16 alone: partial sums (n, splits, channels) fp32 -> means (n, channels) fp16, summed in
// split order (deterministic); for producers that emit the partial sums themselves (pp_conv_own_sums_f16)
extern "C" int pp_channel_mean_finish_f16(const void *partial_ws, void *out, int n, long hw, int channels, int splits, void *stream) {
    if (!partial_ws || !out || n <= 0 || hw <= 0 || channels <= 0 || splits <= 0) return PP_ERR_BAD_ARG;
    const int total = n * channels;
    hipLaunchKernelGGL(k_channel_mean_finish, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(partial_ws), static_cast<__half *>(out), n, channels, splits, 1.0f / (float)hw);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ SE excitation, one launch
// models/layers_transposed.py:289-310 after the squeeze: s = sigmoid(W2 leaky(W1 mean + b1) + b2) per sample.  The torch path is
// five launches per SE block (finish of the channel sums, two GEMMs of a few hundred kFLOP, LeakyReLU, sigmoid) -- 16 blocks per
// forward; here one workgroup per sample does all of it.  The roundings are those of the fp16 torch modules it replaces: the
// mean, the hidden vector (after the bias, again after LeakyReLU) and the gains are rounded to binary16, sums are fp32.
namespace {
__global__ __launch_bounds__(256) void k_se_gains(const float *__restrict__ partial, const __half *__restrict__ mean_in,
                                                  const __half *__restrict__ w1, const __half *__restrict__ b1,
                                                  const __half *__restrict__ w2, const __half *__restrict__ b2, __half *__restrict__ out,
                                                  int c, int hidden, int splits, float inv_hw, float slope) {
    extern __shared__ float se_lds[];      // [c] mean, [hidden] hidden
    float *s_mean = se_lds, *s_hid = se_lds + c;
    const int n = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float m;
        if (partial) {
            float t = 0.f;
            for (int sp = 0; sp < splits; sp++) t += partial[((size_t)n * splits + sp) * c + ch];
            m = __half2float(__float2half_rn(t * inv_hw));
        } else {
            m = __half2float(mean_in[(size_t)n * c + ch]);
        }
        s_mean[ch] = m;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < hidden; j += 4) {   // one wave per hidden unit: dot over the channels
        float t = 0.f;
        for (int ch = lane; ch < c; ch += 64) t += __half2float(w1[(size_t)j * c + ch]) * s_mean[ch];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
        if (lane == 0) {
            float h = __half2float(__float2half_rn(t + __half2float(b1[j])));      // nn.Linear's fp16 output
            h = h > 0.f ? h : h * slope;
            s_hid[j] = __half2float(__float2half_rn(h));                          // LeakyReLU's fp16 output
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float t = 0.f;
        for (int j = 0; j < hidden; j++) t += __half2float(w2[(size_t)ch * hidden + j]) * s_hid[j];
        const float z = __half2float(__float2half_rn(t + __half2float(b2[ch])));
        out[(size_t)n * c + ch] = __float2half_rn(1.0f / (1.0f + expf(-z)));
    }
}
}  // namespace

// gains (n, c) fp16 = sigmoid(W2 leaky(W1 mean + b1) + b2).  Either partial_ws (n, splits, c) fp32 channel SUMS over hw pixels (as
// pp_conv_own_sums_f16 / pp_channel_mean_f16's first pass leave them) or mean (n, c) fp16 is given, the other NULL.
// w1: (hidden, c), b1: (hidden), w2: (c, hidden), b2: (c), all fp16 DEVICE.
extern "C" int pp_se_gains_f16(const void *partial_ws, const void *mean, const void *w1, const void *b1, const void *w2, const void *b2,
                               void *out, int n, long hw, int c, int hidden, int splits, float slope, void *stream) {
    if ((!partial_ws) == (!mean) || !w1 || !b1 || !w2 || !b2 || !out || n <= 0 || c <= 0 || hidden <= 0 || (partial_ws && (hw <= 0 || splits <= 0)))
        return PP_ERR_BAD_ARG;
    if (c > 4096 || hidden > 1024) return PP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_se_gains, dim3((unsigned)n), dim3(256), (size_t)(c + hidden) * sizeof(float), static_cast<hipStream_t>(stream),
                       static_cast<const float *>(partial_ws), static_cast<const __half *>(mean), static_cast<const __half *>(w1),
                       static_cast<const __half *>(b1), static_cast<const __half *>(w2), static_cast<const __half *>(b2),
                       static_cast<__half *>(out), c, hidden, splits, partial_ws ? 1.0f / (float)hw : 0.f, slope);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ NHWC -> planes
// The last head writes its prediction pixel-major (n, h, w, 64: 50 channels + zero padding, see pp_pw_f16); K_A / K_B read
// channel planes (n, 50, h, w) -- the layout of the reference's network output (models/posenet.py:193-202).  A strided
// `.contiguous()` copy does this transposition at about 1 TB/s; here a workgroup moves a tile of 64 pixels x 64 channels through
// LDS: 16-byte reads along the channels, 16-byte writes along the pixels (odd LDS row pitch in dwords: no bank conflicts on the
// transposing side).
namespace {
__global__ __launch_bounds__(256) void k_nhwc64_to_planes(const __half *__restrict__ x, __half *__restrict__ y, long hw, int c_out) {
    __shared__ __half tile[64][66];                    // [channel][pixel], pitch 33 dwords
    const long p0 = (long)blockIdx.x * 64;             // first pixel of the tile inside image blockIdx.y
    const __half *src = x + ((long)blockIdx.y * hw + p0) * 64;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int q = threadIdx.x + r * 256;           // 16-byte chunk: pixel q / 8, channels 8 (q % 8) ..
        const int px = q >> 3, c8 = (q & 7) * 8;
        if (p0 + px < hw) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + (long)px * 64 + c8);
            const __half *h = reinterpret_cast<const __half *>(&v);
#pragma unroll
            for (int e = 0; e < 8; e++) tile[c8 + e][px] = h[e];
        }
    }
    __syncthreads();
    __half *dst = y + (long)blockIdx.y * c_out * hw + p0;
    const bool whole = p0 + 64 <= hw && (hw & 7) == 0;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int q = threadIdx.x + r * 256;           // 16-byte chunk: channel q / 8, pixels 8 (q % 8) ..
        const int c = q >> 3, p8 = (q & 7) * 8;
        if (c >= c_out) continue;
        if (whole) {
            uint4 v;
            __half *h = reinterpret_cast<__half *>(&v);
#pragma unroll
            for (int e = 0; e < 8; e++) h[e] = tile[c][p8 + e];
            *reinterpret_cast<uint4 *>(dst + (long)c * hw + p8) = v;
        } else {
            for (int e = 0; e < 8; e++)
                if (p0 + p8 + e < hw) dst[(long)c * hw + p8 + e] = tile[c][p8 + e];
        }
    }
}
}  // namespace

// x: DEVICE (n, hw, 64) fp16 (an NHWC tensor of 64 channels); y: DEVICE (n, c_out, hw) fp16, c_out <= 64: y[i][c][p] = x[i][p][c].
extern "C" int pp_nhwc64_to_planes_f16(const void *x, void *y, int n, long hw, int c_out, void *stream) {
    if (!x || !y || n <= 0 || hw <= 0 || c_out <= 0 || c_out > 64) return PP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return PP_ERR_BAD_ARG;
    if (n > 65535) return PP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_nhwc64_to_planes, dim3((unsigned)((hw + 63) / 64), (unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __half *>(x), static_cast<__half *>(y), hw, c_out);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ A0 pre-processing
// utils/parse_skeletons.py:52-73 + utils/util.py:44-65 for a batch of equally sized BGR uint8 images (scale 1):
// pad bottom/right to a multiple of `pad_to` with `pad_value`, x / 255 -> float, and emit each image followed by the
// W-mirror of the PADDED image (the reference flips after padding, so the mirror's padding sits on the left).
// out: (2B or B, Hp, Wp, 3) NHWC, fp16 or fp32.  One thread per output pixel of the un-mirrored sample.
namespace {
template <typename OutT>
__device__ __forceinline__ OutT cvt_out(float v);
template <>
__device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <>
__device__ __forceinline__ __half cvt_out<__half>(float v) { return __float2half(v); }

template <typename OutT>
__global__ __launch_bounds__(256) void k_preprocess(const unsigned char *__restrict__ img, OutT *__restrict__ out, int B,
                                                    int H, int W, int Hp, int Wp, int flip, float pad_norm) {
    const long npix = (long)B * Hp * Wp;
    const long stride = (long)gridDim.x * blockDim.x;
    const int ns = flip ? 2 : 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const int x = (int)(i % Wp);
        long t = i / Wp;
        const int y = (int)(t % Hp);
        const long b = t / Hp;
        float v[3];
        if (y < H && x < W) {
            const unsigned char *p = img + ((b * H + y) * W + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = (float)p[c] / 255.0f;  // == np.float32(u8 / 255): checked for all 256 values
        } else {
            v[0] = v[1] = v[2] = pad_norm;
        }
        OutT *o0 = out + (((b * ns) * Hp + y) * Wp + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) o0[c] = cvt_out<OutT>(v[c]);
        if (flip) {
            OutT *o1 = out + (((b * ns + 1) * Hp + y) * Wp + (Wp - 1 - x)) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) o1[c] = cvt_out<OutT>(v[c]);
        }
    }
}
}  // namespace

extern "C" int pp_preprocess_u8(const void *images_u8, void *out, int dtype, int batch, int h, int w, int pad_to,
                                int pad_value, int flip, void *stream) {
    if (!images_u8 || !out || batch <= 0 || h <= 0 || w <= 0 || pad_to <= 0 || (dtype != PP_F16 && dtype != PP_F32))
        return PP_ERR_BAD_ARG;
    const int Hp = (h + pad_to - 1) / pad_to * pad_to, Wp = (w + pad_to - 1) / pad_to * pad_to;
    const long npix = (long)batch * Hp * Wp;
    const dim3 grid(grid_for((npix + 0) / 1)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float pad_norm = (float)pad_value / 255.0f;
    if (dtype == PP_F16)
        hipLaunchKernelGGL(k_preprocess<__half>, grid, block, 0, st, static_cast<const unsigned char *>(images_u8),
                           static_cast<__half *>(out), batch, h, w, Hp, Wp, flip, pad_norm);
    else
        hipLaunchKernelGGL(k_preprocess<float>, grid, block, 0, st, static_cast<const unsigned char *>(images_u8),
                           static_cast<float *>(out), batch, h, w, Hp, Wp, flip, pad_norm);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// The same for a bucket of images of DIFFERENT sizes that share one padded shape (COCO val2017 does: 427x640 and 426x640
// both pad to 448x640).  Each image sits in the top-left corner of its (Hp, Wp) slot of `img`; bytes outside its own
// (h_b, w_b) are never read (the host need not initialise them): the kernel writes pad_value / 255 there, which is what
// padRightDownCorner (utils/util.py:44-65) leaves in the padded image.  sizes: DEVICE int[2][B] = heights then widths.
namespace {
template <typename OutT>
__global__ __launch_bounds__(256) void k_preprocess_ragged(const unsigned char *__restrict__ img, const int *__restrict__ sizes,
                                                           OutT *__restrict__ out, int B, int Hp, int Wp, int flip,
                                                           float pad_norm) {
    const long npix = (long)B * Hp * Wp;
    const long stride = (long)gridDim.x * blockDim.x;
    const int ns = flip ? 2 : 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const int x = (int)(i % Wp);
        long t = i / Wp;
        const int y = (int)(t % Hp);
        const long b = t / Hp;
        const int H = sizes[b], W = sizes[B + b];
        float v[3];
        if (y < H && x < W) {
            const unsigned char *p = img + i * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = (float)p[c] / 255.0f;
        } else {
            v[0] = v[1] = v[2] = pad_norm;
        }
        OutT *o0 = out + (((b * ns) * Hp + y) * Wp + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) o0[c] = cvt_out<OutT>(v[c]);
        if (flip) {
            OutT *o1 = out + (((b * ns + 1) * Hp + y) * Wp + (Wp - 1 - x)) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) o1[c] = cvt_out<OutT>(v[c]);
        }
    }
}
}  // namespace

extern "C" int pp_preprocess_u8_ragged(const void *images_u8, const int *sizes_dev, void *out, int dtype, int batch, int hp,
                                       int wp, int pad_value, int flip, void *stream) {
    if (!images_u8 || !sizes_dev || !out || batch <= 0 || hp <= 0 || wp <= 0 || (dtype != PP_F16 && dtype != PP_F32))
        return PP_ERR_BAD_ARG;
    const long npix = (long)batch * hp * wp;
    const dim3 grid(grid_for(npix)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float pad_norm = (float)pad_value / 255.0f;
    if (dtype == PP_F16)
        hipLaunchKernelGGL(k_preprocess_ragged<__half>, grid, block, 0, st, static_cast<const unsigned char *>(images_u8), sizes_dev,
                           static_cast<__half *>(out), batch, hp, wp, flip, pad_norm);
    else
        hipLaunchKernelGGL(k_preprocess_ragged<float>, grid, block, 0, st, static_cast<const unsigned char *>(images_u8), sizes_dev,
                           static_cast<float *>(out), batch, hp, wp, flip, pad_norm);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ A2 standalone
// predict_refactor's return value (utils/parse_skeletons.py:82-103): heat (h, w, 20) and paf (h, w, 30), HWC float32,
// flip-averaged in the input's dtype.  The fused kernels K_A/K_B never materialise these; this entry point exists for
// callers that want the reference's intermediate arrays.  One thread per (y, x, c).
namespace {
__device__ const signed char e_flip_heat[PP_NUM_HEAT] = {0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16, 18, 19};
__device__ const signed char e_flip_paf[PP_NUM_LIMB] = {0,  2,  1,  4,  3,  6,  5,  8,  7,  12, 13, 14, 9,  10, 11,
                                                       18, 19, 20, 15, 16, 17, 22, 21, 25, 26, 23, 24, 28, 27, 29};
__device__ __forceinline__ float avg_elem(const __half *p, long i0, long i1, bool flip) {
    return flip ? __half2float(__hmul(__hadd(p[i0], p[i1]), __float2half(0.5f))) : __half2float(p[i0]);
}
__device__ __forceinline__ float avg_elem(const float *p, long i0, long i1, bool flip) {
    return flip ? __fadd_rn(p[i0], p[i1]) / 2.0f : p[i0];
}
template <typename T>
__global__ __launch_bounds__(256) void k_flip_average(const T *__restrict__ net, int batch, int h, int w, int flip,
                                                      float *__restrict__ heat, float *__restrict__ paf) {
    const long plane = (long)h * w;
    const long total = (long)batch * plane * PP_NUM_CH;
    const long stride = (long)gridDim.x * blockDim.x;
    const int ns = flip ? 2 : 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % PP_NUM_CH);
        long t = i / PP_NUM_CH;
        const int x = (int)(t % w);
        t /= w;
        const int y = (int)(t % h);
        const long b = t / h;
        const int cf = c < PP_NUM_LIMB ? e_flip_paf[c] : PP_NUM_LIMB + e_flip_heat[c - PP_NUM_LIMB];
        const long i0 = ((b * ns) * PP_NUM_CH + c) * plane + (long)y * w + x;
        const long i1 = ((b * ns + 1) * PP_NUM_CH + cf) * plane + (long)y * w + (w - 1 - x);
        const float v = avg_elem(net, i0, i1, flip != 0);
        if (c < PP_NUM_LIMB) paf[((b * h + y) * w + x) * PP_NUM_LIMB + c] = v;
        else heat[((b * h + y) * w + x) * PP_NUM_HEAT + (c - PP_NUM_LIMB)] = v;
    }
}
}  // namespace

extern "C" int pp_flip_average(const void *net_out_dev, int dtype, int batch, int h, int w, int flip, float *heat_hwc_dev,
                               float *paf_hwc_dev, void *stream) {
    if (!net_out_dev || !heat_hwc_dev || !paf_hwc_dev || batch <= 0 || h <= 0 || w <= 0 || (dtype != PP_F16 && dtype != PP_F32))
        return PP_ERR_BAD_ARG;
    const long total = (long)batch * h * w * PP_NUM_CH;
    const dim3 grid(grid_for(total)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == PP_F16)
        hipLaunchKernelGGL(k_flip_average<__half>, grid, block, 0, st, static_cast<const __half *>(net_out_dev), batch, h, w,
                           flip, heat_hwc_dev, paf_hwc_dev);
    else
        hipLaunchKernelGGL(k_flip_average<float>, grid, block, 0, st, static_cast<const float *>(net_out_dev), batch, h, w,
                           flip, heat_hwc_dev, paf_hwc_dev);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------ A0, scale != 1
// cv2.resize(image, fx=fy=scale, INTER_CUBIC) on the uint8 image (utils/parse_skeletons.py:204): OpenCV's fixed-point
// bicubic restated (see oracle/posepaf_oracle.c, orc_resize_cubic_u8; parity unpinned).  One thread per output pixel.
namespace {
__device__ __forceinline__ void cubic_coeffs_i(float x, int c[4]) {
    const float A = -0.75f;
    const float xp = __fadd_rn(x, 1.0f), y = __fadd_rn(1.0f, -x);
    const float c0 = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(A, xp), 3.75f), xp), -6.0f), xp), 3.0f);
    const float c1 = __fadd_rn(__fmul_rn(__fmul_rn(__fadd_rn(__fmul_rn(1.25f, x), -2.25f), x), x), 1.0f);
    const float c2 = __fadd_rn(__fmul_rn(__fmul_rn(__fadd_rn(__fmul_rn(1.25f, y), -2.25f), y), y), 1.0f);
    const float c3 = __fadd_rn(__fadd_rn(__fadd_rn(1.0f, -c0), -c1), -c2);
    c[0] = __float2int_rn(__fmul_rn(c0, 2048.0f));
    c[1] = __float2int_rn(__fmul_rn(c1, 2048.0f));
    c[2] = __float2int_rn(__fmul_rn(c2, 2048.0f));
    c[3] = __float2int_rn(__fmul_rn(c3, 2048.0f));
}
__device__ __forceinline__ int clampi_e(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void k_resize_u8(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, int B,
                                                   int sh, int sw, int dh, int dw, double scale_x, double scale_y) {
    const long total = (long)B * dh * dw;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int dx = (int)(i % dw);
        long t = i / dw;
        const int dy = (int)(t % dh);
        const long b = t / dh;
        float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
        const int sx = (int)floorf(fx);
        fx = __fadd_rn(fx, -(float)sx);
        float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy = __fadd_rn(fy, -(float)sy);
        int ia[4], ib[4];
        cubic_coeffs_i(fx, ia);
        cubic_coeffs_i(fy, ib);
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 4; ky++) {
            const unsigned char *row = src + ((b * sh + clampi_e(sy - 1 + ky, 0, sh - 1)) * (long)sw) * 3;
            int hs[3] = {0, 0, 0};
#pragma unroll
            for (int kx = 0; kx < 4; kx++) {
                const unsigned char *p = row + (long)clampi_e(sx - 1 + kx, 0, sw - 1) * 3;
                hs[0] += (int)p[0] * ia[kx];
                hs[1] += (int)p[1] * ia[kx];
                hs[2] += (int)p[2] * ia[kx];
            }
            acc[0] += hs[0] * ib[ky];
            acc[1] += hs[1] * ib[ky];
            acc[2] += hs[2] * ib[ky];
        }
        unsigned char *o = dst + i * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int v = (acc[c] + (1 << 21)) >> 22;
            o[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}
}  // namespace

extern "C" int pp_resize_u8_cubic(const void *src, void *dst, int batch, int sh, int sw, int dh, int dw, double scale_x,
                                  double scale_y, void *stream) {
    if (!src || !dst || batch <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return PP_ERR_BAD_ARG;
    const long total = (long)batch * dh * dw;
    hipLaunchKernelGGL(k_resize_u8, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char *>(src), static_cast<unsigned char *>(dst), batch, sh, sw, dh, dw, scale_x,
                       scale_y);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}
