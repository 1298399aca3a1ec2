// posepaf_pwconv.hip -- fused point-wise (1x1) convolution for the IMHN forward on gfx950 matrix cores:
//
//     Y[m][n] = act( sum_k X[m][k] * W[n][k] + bias[n] (+ R[m][n]) ) (+ P[m][n])        fp16 in/out, fp32 accumulate
//
// X is the channels-last activation viewed as [M = batch*H*W][K], W the conv weight [N][K] (a (Cout, Cin, 1, 1) tensor),
// R the residual added BEFORE the activation (bottleneck skip), P a tensor added AFTER it.  Two thirds of the IMHN's
// convolutions are 1x1 and HBM-bound (256->128 and 128->256 on a 64x128x128 batch move 0.8 GB each); run through
// MIOpen they cost a convolution pass plus a separate bias/activation/residual pass over the output.  Here the
// epilogue rides on the GEMM and the activation is written once.
//
// Structure (wave64, v_mfma_f32_32x32x16_f16):
//  * persistent grid (one 256-thread workgroup per CU); the whole weight matrix sits in LDS for the kernel's lifetime
//    (rows padded by 16 B: the B-fragment read `W[n0 + (lane&31)][k0 + 8*(lane>>5) ..+8]` is then bank-conflict free);
//  * each wave owns 32 consecutive rows of X per tile: its A fragments (lane: 8 consecutive k of row lane&31 = one
//    16-byte global load) are loaded ONCE into registers and re-used for every 128-column chunk of N;
//  * the 32x32 accumulator tiles (column on the lane, 16 rows in registers) are transposed through a small LDS stage
//    so that residual reads and output writes are full 16-byte row-major vectors.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/posepaf.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kRowsPerWave = 32;
constexpr int kBM = kWaves * kRowsPerWave;  // 128 rows per workgroup tile
constexpr int kNChunk = 128;                // accumulator columns held at once: 4 tiles x 16 regs
constexpr int kStageCols = 64;              // epilogue transposition stage: 32 rows x 64 cols per wave

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// LDS: [W: N rows x (K + 8) halves][out stage: 4 waves x 32 x (64 + 8) halves][X slabs: 4 waves x 32 x (K + 8) halves]
template <int KSTEPS>  // K / 16
__global__ __launch_bounds__(kThreads, 1) void k_pwconv(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                        const _Float16 *__restrict__ bias, const _Float16 *__restrict__ R,
                                                        const _Float16 *__restrict__ P, _Float16 *__restrict__ Y, long M,
                                                        int N, float slope, int has_act) {
    constexpr int K = KSTEPS * 16;
    constexpr int WLD = K + 8;  // padded leading dimension (halves)
    constexpr int SLD = kStageCols + 8;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    _Float16 *sW = reinterpret_cast<_Float16 *>(lds_raw);
    _Float16 *sStage = sW + (size_t)N * WLD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;

    // ---- weights -> LDS, once per workgroup (16-byte vectors)
    {
        const int vec_per_row = K / 8;
        const int nvec = N * vec_per_row;
        for (int v = threadIdx.x; v < nvec; v += kThreads) {
            const int n = v / vec_per_row, kv = v - n * vec_per_row;
            *reinterpret_cast<uint4 *>(sW + (size_t)n * WLD + kv * 8) = *reinterpret_cast<const uint4 *>(W + (size_t)n * K + kv * 8);
        }
    }
    __syncthreads();

    _Float16 *stage = sStage + (size_t)wave * kRowsPerWave * SLD;
    const long ntiles = (M + kBM - 1) / kBM;
    // X rows reach the wave as FULL lines: lane l of a load instruction takes 16 consecutive bytes, a wave instruction
    // 1 KB of consecutive row data (fragment-shaped loads -- 32 rows x 32 B per instruction -- touch 4x as many lines
    // per instruction and ran at 2.9 TB/s).  The rows are parked in this wave's LDS slab (rows padded by 16 B) and the
    // MFMA A fragments are read back from there with ds_read_b128.  Software pipeline: the global loads of tile t+1
    // are issued before tile t's MFMAs and written to LDS after tile t's fragments have been read.
    constexpr int XLD = K + 8;
    constexpr int kVecPerRow = K / 8;                       // 16-byte vectors per row
    constexpr int kLoads = kRowsPerWave * kVecPerRow / 64;  // per lane per tile (K/16)
    _Float16 *sX = sStage + (size_t)kWaves * kRowsPerWave * SLD + (size_t)wave * kRowsPerWave * XLD;
    uint4 xg[kLoads];
    auto load_x = [&](long tile) {
        const long base_row = tile * kBM + (long)wave * kRowsPerWave;
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const int v = i * 64 + lane;
            const int rr = v / kVecPerRow, cv = v - rr * kVecPerRow;
            const long grow = base_row + rr;
            if (tile < ntiles && grow < M) xg[i] = *reinterpret_cast<const uint4 *>(X + grow * K + cv * 8);
            else xg[i] = make_uint4(0, 0, 0, 0);
        }
    };
    auto park_x = [&]() {
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const int v = i * 64 + lane;
            const int rr = v / kVecPerRow, cv = v - rr * kVecPerRow;
            *reinterpret_cast<uint4 *>(sX + rr * XLD + cv * 8) = xg[i];
        }
    };
    load_x(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * kBM + (long)wave * kRowsPerWave;
        park_x();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        half8 a[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) a[s] = *reinterpret_cast<const half8 *>(sX + r * XLD + s * 16 + hh * 8);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        load_x(tile + gridDim.x);  // in flight during this tile's MFMAs and stores
        for (int n0 = 0; n0 < N; n0 += kNChunk) {
            const int ncols = (N - n0) < kNChunk ? (N - n0) : kNChunk;
            const int ntile = ncols / 32;
            float16v acc[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (t < ntile) {
                        // B fragment: B[k = 16 s + 8 hh + j][col r] = W[n0 + 32 t + r][k]
                        const half8 b = *reinterpret_cast<const half8 *>(sW + (size_t)(n0 + 32 * t + r) * WLD + s * 16 + hh * 8);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b, acc[t], 0, 0, 0);
                    }
                }
            }
            // ---- epilogue, 64 columns (two 32x32 tiles) at a time through the wave's LDS stage
#pragma unroll
            for (int t0 = 0; t0 < 4; t0 += 2) {
                if (t0 >= ntile) break;
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    const int t = t0 + tt;
                    if (t < ntile) {
                        const float bv = (float)bias[n0 + 32 * t + r];
#pragma unroll
                        for (int q = 0; q < 16; q++) {
                            // C/D map: col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
                            const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
                            stage[row * SLD + tt * 32 + r] = (_Float16)(acc[t0 + tt][q] + bv);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int cols = (ntile - t0) >= 2 ? 64 : 32;
                const int vec_per_row = cols / 8;              // 8 or 4 sixteen-byte vectors per row
                const int rows_per_pass = 64 / vec_per_row;    // 8 or 16 rows per wave instruction
                for (int rb = 0; rb < kRowsPerWave; rb += rows_per_pass) {
                    const int rr = rb + lane / vec_per_row, cv = lane % vec_per_row;
                    const long grow = row0 + rr;
                    if (grow < M) {
                        half8 v = *reinterpret_cast<const half8 *>(stage + rr * SLD + cv * 8);
                        const long goff = grow * N + n0 + 32 * t0 + cv * 8;
                        float f[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) f[j] = (float)v[j];
                        if (R) {
                            const half8 rv = *reinterpret_cast<const half8 *>(R + goff);
#pragma unroll
                            for (int j = 0; j < 8; j++) f[j] += (float)rv[j];
                        }
                        if (has_act) {
#pragma unroll
                            for (int j = 0; j < 8; j++) f[j] = leaky(f[j], slope);
                        }
                        if (P) {
                            const half8 pv = *reinterpret_cast<const half8 *>(P + goff);
#pragma unroll
                            for (int j = 0; j < 8; j++) f[j] += (float)pv[j];
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) v[j] = (_Float16)f[j];
                        *reinterpret_cast<half8 *>(Y + goff) = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

template <int KSTEPS>
hipError_t launch(const void *x, const void *w, const void *bias, const void *res, const void *post, void *y, long M, int N,
                  float slope, int has_act, int n_cu, hipStream_t st) {
    constexpr int K = KSTEPS * 16;
    const size_t lds = (size_t)N * (K + 8) * 2 + (size_t)kWaves * kRowsPerWave * (kStageCols + 8) * 2 +
                       (size_t)kWaves * kRowsPerWave * (K + 8) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pwconv<KSTEPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    long tiles = (M + kBM - 1) / kBM;
    int grid = (int)(tiles < n_cu ? tiles : n_cu);
    hipLaunchKernelGGL(k_pwconv<KSTEPS>, dim3(grid), dim3(kThreads), lds, st, static_cast<const _Float16 *>(x),
                       static_cast<const _Float16 *>(w), static_cast<const _Float16 *>(bias),
                       static_cast<const _Float16 *>(res), static_cast<const _Float16 *>(post), static_cast<_Float16 *>(y), M, N,
                       slope, has_act);
    return hipGetLastError();
}

}  // namespace

// 1 if (K, N) is a shape the kernel takes: K in {64, 128, 192, 256}, N % 32 == 0, weights + stage fit the 160 KB LDS
extern "C" int pp_pwconv_supported(int K, int N) {
    if (!(K == 64 || K == 128 || K == 192 || K == 256) || N <= 0 || (N & 31)) return 0;
    const size_t lds = (size_t)N * (K + 8) * 2 + (size_t)kWaves * kRowsPerWave * (kStageCols + 8) * 2 +
                       (size_t)kWaves * kRowsPerWave * (K + 8) * 2;
    return lds <= 160 * 1024 - 2048 ? 1 : 0;
}

extern "C" int pp_pwconv_f16(const void *x, const void *w, const void *bias, const void *residual, const void *post, void *y,
                             long M, int K, int N, float slope, int has_act, void *stream) {
    if (!x || !w || !bias || !y || M <= 0 || !pp_pwconv_supported(K, N)) return PP_ERR_BAD_ARG;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(post);
    if (al & 15) return PP_ERR_BAD_ARG;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return PP_ERR_NO_DEVICE;
        n_cu = prop.multiProcessorCount;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e;
    switch (K) {
        case 64: e = launch<4>(x, w, bias, residual, post, y, M, N, slope, has_act, n_cu, st); break;
        case 128: e = launch<8>(x, w, bias, residual, post, y, M, N, slope, has_act, n_cu, st); break;
        case 192: e = launch<12>(x, w, bias, residual, post, y, M, N, slope, has_act, n_cu, st); break;
        default: e = launch<16>(x, w, bias, residual, post, y, M, N, slope, has_act, n_cu, st); break;
    }
    return e == hipSuccess ? PP_OK : PP_ERR_HIP;
}
