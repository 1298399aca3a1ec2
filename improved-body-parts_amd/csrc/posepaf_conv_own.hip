// posepaf_conv_own.hip -- A1 forward: hand-written implicit-GEMM convolution for gfx950 (CDNA4) on the matrix cores,
//   y = leaky(conv(x, w) + bias (+ residual)) (+ post)      stride 1, square kernel (1x1, 3x3, dilated), fp16 in / out,
// fp32 accumulate -- Conv2d + folded BatchNorm + LeakyReLU (+ the residual / hourglass adds) of models/layers_transposed.py
// (Conv :90-122, Residual :12-48, Hourglass :199-286), in ONE kernel and without composable_kernel.
//
// GEMM view: rows = output pixels (N*Ho*Wo, NHWC so a pixel's channels are contiguous), columns = output channels,
// K = taps x input channels walked tap by tap in steps of 64 channels, so that every K-step of a pixel row is ONE
// contiguous 128-byte run of the input (or the zero page when the tap falls outside the image).
//
// Workgroup tile 256 pixels x BN channels (BN = 256 / 128 / 64: the largest that divides C_out), 512 threads = 8 waves
// of 64 lanes, each wave owns a (256 / WM) x 64 block of the output with v_mfma_f32_16x16x32_f16 (fp32 accumulators in
// registers: 128 / 64 / 32 VGPRs).  Operand tiles travel HBM -> LDS by direct LDS-DMA loads (global_load_lds_dwordx4,
// 16 B per lane, no VGPR staging), double buffered: the loads of K-step k+1 are in flight while K-step k is multiplied.
// LDS image: 1-KiB sub-tiles of 16 rows x 32 halves; a wave's DMA instruction fills exactly one sub-tile (lane-linear
// destination), the XOR swizzle that keeps the 16-byte fragment reads (ds_read_b128) spread over the banks is applied to
// the per-lane SOURCE address and to the read address (cdna_hip_programming.md T2).
// The MFMA takes the WEIGHT fragment as its A operand and the PIXEL fragment as B, so a lane's four accumulator registers
// are four CONSECUTIVE output channels of one pixel: the epilogue (bias, residual, LeakyReLU, post add, fp16 pack) stores
// 8 contiguous bytes per lane straight from registers, no LDS transpose.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/posepaf.h"

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 256;      // output pixels per workgroup
constexpr int NTHREADS = 512;
constexpr int SUB = 1024;    // bytes of one LDS sub-tile: 16 rows x 32 halves

struct ConvParams {
    const _Float16 *x, *w, *bias, *extra, *extra2;   // extra2: a second tensor added after the activation (mode 3; 3x3 halo kernel only)
    _Float16 *y, *y2;  // y2: second output of mode 4
    const void *zero;  // >= 16 zero bytes: source of every tap that falls outside the image
    int N, H, W, C, K, R, pad, dil, Ho, Wo;
    int pad_y, pad_x;  // implicit-GEMM kernel: tap (r, s) reads input pixel (oy + r * dil - pad_y, ox + s * dil - pad_x); = pad, except
                       // for the collapsed upsample convolution below
    int up_out, py, px;// up_out = 1: the kernel's (n, oy, ox) output pixel is pixel (2 oy + py, 2 ox + px) of y / extra / extra2,
                       // tensors of twice the size (one output phase of a convolution behind a x2 nearest upsample)
    float *csum;       // NULL, or per-wave partial channel sums of y (halo kernel): [image][tile][4 pixel waves][K] floats -- the SE
                       // block's squeeze (models/layers_transposed.py:298-303) without a pass of its own over y
    int ldy;           // elements between consecutive pixels of y (K: packed; larger: y is a channel slice of a wider tensor,
                       // e.g. one half of the backbone's concatenation, models/layers_transposed.py:193-195)
    int ldx = 0;       // the same for x (3x3 halo kernel only); 0: packed (C)
    long M;            // N * Ho * Wo
    int mode;          // 0 none, 1 extra added before the activation, 2 after, 3 extra AND extra2 after (3x3 halo kernel),
                       // 4 = mode 1 plus a SECOND OUTPUT y2 = y + extra2 (every kernel)
    int up;            // 1: x is the input at HALF resolution (N x H/2 x W/2 x C); the kernel reads pixel (y >> 1, x >> 1): the x2
                       // nearest upsample in front of the convolution costs no pass of its own (3x3 halo kernel only)
    float slope;
    int stagger;       // persistent 3x3 kernel: start delay of the last workgroup of an XCD, in units of s_sleep(127) (~8100 cycles)
    int dbg;           // ablation switches (POSEPAF_CONV_DBG, diagnostics only): 1 no DMA in the loop, 2 no MFMA, 4 no fragment
                       // reads, 8 no epilogue stores.  0 in production.
};

__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned char *lds_wave_base) {
    // 64 lanes x 16 B: lane i's bytes land at lds_wave_base + 16 * i (the destination is wave-uniform, M0-based)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// The same through a buffer resource (buffer_load_dwordx4 ... offen lds): the address is base(rsrc) + voffset (per lane) + soffset
// (scalar), so a load whose lane part is fixed costs NO address arithmetic, and a lane whose voffset lies outside num_records
// gets ZEROS written to LDS (probed on gfx950: tools/buffer_lds_probe.hip) -- the zero padding of the convolution for free.
__device__ __forceinline__ void lds_dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset, unsigned char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, (int)voffset, (int)soffset, 0, 0);
}

// Epilogue from registers with 16-byte stores.  After the MFMAs a lane holds, per (pixel tile i, channel tile j), four
// consecutive channels (4 * (lane >> 4) ...) of pixel lane & 15: 8 bytes.  A row-per-lane epilogue of 8-byte stores is
// store-ISSUE bound (32 of them per lane here).  Lanes l and l ^ 16 hold ADJACENT channel quads of the same pixel, so channel
// tiles are handled in pairs: the even-quad lane takes both quads of tile j, the odd-quad lane both quads of tile j + 1 (one
// cross-lane exchange of four accumulators each way), and every lane then finishes eight consecutive channels -- bias,
// residual, LeakyReLU, post add in fp32, one 16-byte load of `extra`, one 16-byte store: half the memory instructions.
// pixel_of(i): flat output pixel index of this lane in pixel tile i, or -1 (outside the tensor).
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// sums_row (SUMS): where this wave's partial channel sums go (float[K] of its (image, tile, pixel wave)): the binary16 outputs are
// summed in fp32 over the wave's pixels -- per lane over its pixel tiles, then over the 16 lanes that hold the same channels.
template <int MODE, int PT, int CT, bool SUMS = false, typename PixelOf>
__device__ __forceinline__ void epilogue_body(const float4_t (&acc)[PT][CT], const ConvParams &p, int lane, int nbase,
                                              PixelOf pixel_of, bool do_store, float *sums_row = nullptr) {
    static_assert(CT % 2 == 0, "channel tiles are finished in pairs");
    const int g = lane >> 4, odd = g & 1, cbase = (g & ~1) * 4;
    const float2_t slope2 = float2_t{p.slope, p.slope};
#pragma unroll
    for (int jp = 0; jp < CT; jp += 2) {
        float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int co = nbase + (jp + odd) * 16 + cbase;
        const half8_t bv = *reinterpret_cast<const half8_t *>(p.bias + co);
        float2_t b2[4];
#pragma unroll
        for (int e = 0; e < 4; e++) b2[e] = float2_t{(float)bv[2 * e], (float)bv[2 * e + 1]};
#pragma unroll
        for (int i = 0; i < PT; i++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                // v_permlane16_swap: the odd 16-lane rows of the first operand change places with the even rows of the second.
                // Afterwards the first holds, in even rows, the own quad of tile jp and, in odd rows, the lower neighbour's quad of
                // tile jp + 1; the second the upper neighbour's quad of tile jp (even rows) and the own quad of tile jp + 1 (odd).
                const float a0 = acc[i][jp][e], a1 = acc[i][jp + 1][e];   // (bit_cast straight from the vector element miscompiles)
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
                v[e] = __builtin_bit_cast(float, (unsigned)sw[0]);
                v[4 + e] = __builtin_bit_cast(float, (unsigned)sw[1]);
            }
            const long m = pixel_of(i);
            if (m >= 0) {
                const long o = m * p.K + co;
                half8_t ev = half8_t{0, 0, 0, 0, 0, 0, 0, 0}, ev2 = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (MODE) ev = *reinterpret_cast<const half8_t *>(p.extra + o);
                if (MODE >= 3) ev2 = *reinterpret_cast<const half8_t *>(p.extra2 + o);
                half8_t out, out2;
#pragma unroll
                for (int e = 0; e < 4; e++) {   // two channels at a time: v_pk_add_f32 / v_pk_mul_f32
                    float2_t t = float2_t{v[2 * e], v[2 * e + 1]} + b2[e];
                    const float2_t x = float2_t{(float)ev[2 * e], (float)ev[2 * e + 1]};
                    if (MODE == 1 || MODE == 4) t += x;
                    const float2_t u = t * slope2;   // LeakyReLU for 0 <= slope <= 1 (checked by the launcher); slope 1: none
                    t = float2_t{fmaxf(t[0], u[0]), fmaxf(t[1], u[1])};
                    if (MODE == 2) t += x;
                    // mode 3 = the three-way add (k_add3) of the fp16 convolution output it replaces: the activation is rounded
                    // to binary16 first, the sum of the three is formed in fp32 and rounded once
                    if (MODE == 3)
                        t = float2_t{(float)(_Float16)t[0], (float)(_Float16)t[1]} + x + float2_t{(float)ev2[2 * e], (float)ev2[2 * e + 1]};
                    out[2 * e] = (_Float16)t[0];
                    out[2 * e + 1] = (_Float16)t[1];
                    if (MODE == 4) {   // the tensor add y + extra2 of the two binary16 tensors: exact sum, rounded once
                        out2[2 * e] = (_Float16)((float)out[2 * e] + (float)ev2[2 * e]);
                        out2[2 * e + 1] = (_Float16)((float)out[2 * e + 1] + (float)ev2[2 * e + 1]);
                    }
                }
                if (do_store) *reinterpret_cast<half8_t *>(p.y + (m * p.ldy + co)) = out;
                if (MODE == 4 && do_store) *reinterpret_cast<half8_t *>(p.y2 + o) = out2;
                if (SUMS) {
#pragma unroll
                    for (int e = 0; e < 8; e++) csum[e] += (float)out[e];
                }
            }
        }
        if (SUMS) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) csum[e] += __shfl_xor(csum[e], d);
            }
            if ((lane & 15) == 0) {
                float4 *dst = reinterpret_cast<float4 *>(sums_row + co);
                dst[0] = make_float4(csum[0], csum[1], csum[2], csum[3]);
                dst[1] = make_float4(csum[4], csum[5], csum[6], csum[7]);
            }
        }
    }
}

// The same epilogue with the `extra` / `extra2` vectors ALREADY in registers (k_pw requests them before its MFMAs, so their
// latency hides under the matrix work instead of following it): pre[i][jp / 2] / pre2[...] = the 16 bytes this lane adds to
// pixel tile i, channel-tile pair jp.  Same arithmetic, same stores as epilogue_body.
// POOL: the group is a 2-row block (tiles 0 .. PT/2-1 = row y, tiles PT/2 .. = row y + 1, same columns) and the 2x2 max-pool of
// the tensor just produced (y, or y2 in mode 4) goes out as well -- `low = lv[1](pool(x))` of the hourglass
// (models/layers_transposed.py:262-266) finds its input without a pooling pass: vertical pairs are two registers of one lane,
// horizontal pairs are neighbouring lanes (same channels).  pool_base: index of the group's first pooled pixel.
__device__ __forceinline__ half8_t lane_xor1(const half8_t &v) {
    union {
        half8_t h;
        int u[4];
    } a, b;
    a.h = v;
#pragma unroll
    for (int k = 0; k < 4; k++) b.u[k] = __shfl_xor(a.u[k], 1);
    return b.h;
}

template <int MODE, int PT, int CT, bool POOL, typename PixelOf>
__device__ __forceinline__ void epilogue_preloaded(const float4_t (&acc)[PT][CT], const ConvParams &p, int lane, int nbase,
                                                   PixelOf pixel_of, const half8_t (&pre)[PT][CT / 2], const half8_t (&pre2)[PT][CT / 2],
                                                   _Float16 *pool = nullptr, long pool_base = 0) {
    const int g = lane >> 4, odd = g & 1, cbase = (g & ~1) * 4;
    const float2_t slope2 = float2_t{p.slope, p.slope};
#pragma unroll
    for (int jp = 0; jp < CT; jp += 2) {
        const int co = nbase + (jp + odd) * 16 + cbase;
        const half8_t bv = *reinterpret_cast<const half8_t *>(p.bias + co);
        float2_t b2[4];
#pragma unroll
        for (int e = 0; e < 4; e++) b2[e] = float2_t{(float)bv[2 * e], (float)bv[2 * e + 1]};
        half8_t keep[PT];
#pragma unroll
        for (int i = 0; i < PT; i++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float a0 = acc[i][jp][e], a1 = acc[i][jp + 1][e];
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
                v[e] = __builtin_bit_cast(float, (unsigned)sw[0]);
                v[4 + e] = __builtin_bit_cast(float, (unsigned)sw[1]);
            }
            const long m = pixel_of(i);
            keep[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (m >= 0) {
                const half8_t ev = pre[i][jp / 2], ev2 = pre2[i][jp / 2];
                half8_t out, out2;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float2_t t = float2_t{v[2 * e], v[2 * e + 1]} + b2[e];
                    const float2_t x = float2_t{(float)ev[2 * e], (float)ev[2 * e + 1]};
                    if (MODE == 1 || MODE == 4) t += x;
                    const float2_t u = t * slope2;
                    t = float2_t{fmaxf(t[0], u[0]), fmaxf(t[1], u[1])};
                    if (MODE == 2) t += x;
                    out[2 * e] = (_Float16)t[0];
                    out[2 * e + 1] = (_Float16)t[1];
                    if (MODE == 4 || MODE == 5) {
                        out2[2 * e] = (_Float16)((float)out[2 * e] + (float)ev2[2 * e]);
                        out2[2 * e + 1] = (_Float16)((float)out[2 * e + 1] + (float)ev2[2 * e + 1]);
                    }
                }
                *reinterpret_cast<half8_t *>(p.y + (m * p.ldy + co)) = out;
                if (MODE == 4 || MODE == 5) *reinterpret_cast<half8_t *>(p.y2 + (m * p.K + co)) = out2;
                keep[i] = (MODE == 4 || MODE == 5) ? out2 : out;
            }
        }
        if (POOL && PT >= 2) {
            constexpr int CB = PT >= 2 ? PT / 2 : 1;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const half8_t vert = __builtin_elementwise_max(keep[cb], keep[CB + cb]);
                const half8_t pm = __builtin_elementwise_max(vert, lane_xor1(vert));
                if (!(lane & 1)) *reinterpret_cast<half8_t *>(pool + ((pool_base + cb * 8 + ((lane & 15) >> 1)) * p.K + co)) = pm;
            }
        }
    }
}

template <int PT, int CT, typename PixelOf>
__device__ __forceinline__ void epilogue_store(const float4_t (&acc)[PT][CT], const ConvParams &p, int lane, int nbase,
                                               PixelOf pixel_of, bool do_store, float *sums_row = nullptr) {
    if (nbase >= p.K) return;   // a wave whose 64 channels lie past C_out (halo kernel, C_out = 64 mod 128): wave-uniform
    // the residual mode is wave-uniform: three bodies, one scalar branch, no per-element selects
    if (sums_row) epilogue_body<0, PT, CT, true>(acc, p, lane, nbase, pixel_of, do_store, sums_row);   // mode 0 only (launcher)
    else if (p.mode == 0) epilogue_body<0, PT, CT>(acc, p, lane, nbase, pixel_of, do_store);
    else if (p.mode == 1) epilogue_body<1, PT, CT>(acc, p, lane, nbase, pixel_of, do_store);
    else if (p.mode == 2) epilogue_body<2, PT, CT>(acc, p, lane, nbase, pixel_of, do_store);
    else if (p.mode == 3) epilogue_body<3, PT, CT>(acc, p, lane, nbase, pixel_of, do_store);
    else epilogue_body<4, PT, CT>(acc, p, lane, nbase, pixel_of, do_store);
}

// DIAGNOSTIC (MASK bit 64 of k_conv3x3_halo): the epilogue's arithmetic and store COUNT with every store instruction covering
// 8 pixels x 128 B (full cache lines) instead of 16 pixels x 64 B; the values land at wrong places.  Times the store pattern only.
template <int PT, int CT, typename PixelOfQ>
__device__ __forceinline__ void epilogue_fullline_probe(const float4_t (&acc)[PT][CT], const ConvParams &p, int lane, int nbase,
                                                        PixelOfQ pixel_of_q) {
    const int co = nbase + (lane >> 3) * 8;
    const half8_t bv = *reinterpret_cast<const half8_t *>(p.bias + co);
#pragma unroll
    for (int jp = 0; jp < CT; jp += 2)
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const long o = pixel_of_q(i * 16 + (jp >> 1) * 8 + (lane & 7)) * p.K + co;
            half8_t out;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float t = (e < 4 ? acc[i][jp][e] : acc[i][jp + 1][e - 4]) + (float)bv[e];
                t = t > 0.f ? t : t * p.slope;
                out[e] = (_Float16)t;
            }
            *reinterpret_cast<half8_t *>(p.y + o) = out;
        }
}

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// Hand-placed matrix instruction and LDS fragment read (see k_conv3x3_halo): the accumulator is tied to its registers.
__device__ __forceinline__ void mfma_acc(float4_t &c, const half8_t &a, const half8_t &b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void lds_read16(half8_t &dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K loop in PHASES of 32 input channels.  Four phase buffers in LDS (pixel slice 256 x 32 + weight slice BN x 32 halves each);
// while phase p is multiplied, the LDS-DMA loads of phases p+1 and p+2 are in flight and those of phase p+3 are issued into the
// buffer phase p-1 used (every wave is past its phase p-1 reads once it has left the barrier that opens phase p).  One raw
// s_barrier per phase, counted vmcnt (never 0 inside the loop): the DMA stays in flight ACROSS barriers.
template <int BN>
__global__ __launch_bounds__(NTHREADS) void k_conv_igemm(const ConvParams p) {
    constexpr int WN = BN / 64;          // waves along the output channels
    constexpr int WM = 8 / WN;           // waves along the pixels
    constexpr int PM = BM / WM;          // pixels per wave
    constexpr int PT = PM / 16;          // 16-pixel tiles per wave
    constexpr int CT = 4;                // 16-channel tiles per wave
    constexpr int A_BYTES = BM * 32 * 2;         // pixel slice of one phase: 16 sub-tiles
    constexpr int B_BYTES = BN * 32 * 2;         // weight slice of one phase: BN / 16 sub-tiles
    constexpr int PBUF = A_BYTES + B_BYTES;
    constexpr int NBUF = 4;
    constexpr int BL = BN == 256 ? 2 : 1;        // weight sub-tiles each wave stages per phase (BN = 64: waves 4..7 repeat 0..3)
    constexpr int NL = 2 + BL;                   // DMA instructions per wave and phase
    extern __shared__ __align__(16) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- this lane's place in a DMA'd sub-tile (swizzle on the source side)
    const int b = lane * 16;
    const int bs = b ^ (((b >> 9) & 1) << 5);
    const int row_in = bs >> 6;   // 0..15
    const int kbyte = bs & 63;    // byte offset inside the 32-half k slice
    // pixel rows this lane fetches: row blocks 2*wave and 2*wave+1
    int oy[2], ox[2];
    long xoff[2];
    bool rowok[2];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const long m = m0 + (wave * 2 + i) * 16 + row_in;
        rowok[i] = m < p.M;
        const long mm = rowok[i] ? m : 0;
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (long)n * HoWo);
        oy[i] = rem / p.Wo;
        ox[i] = rem - oy[i] * p.Wo;
        xoff[i] = (((long)n * p.H + oy[i]) * p.W + ox[i]) * p.C * 2 + kbyte;
    }
    // weight rows this lane fetches
    long woff[BL];
    int wsub[BL];
    const long wrow = (long)p.R * p.R * p.C * 2;  // bytes per output channel
#pragma unroll
    for (int j = 0; j < BL; j++) {
        wsub[j] = BN == 256 ? wave * 2 + j : (BN == 128 ? wave : (wave & 3));
        woff[j] = (long)(n0 + wsub[j] * 16 + row_in) * wrow + kbyte;
    }
    const int kc2 = p.C / 32;           // phases per tap
    const int np = p.R * p.R * kc2;     // phases in all
    const char *xb = reinterpret_cast<const char *>(p.x);
    const char *wb = reinterpret_cast<const char *>(p.w);
    const char *zp = reinterpret_cast<const char *>(p.zero);

    int st_r = 0, st_s = 0, st_c = 0;  // tap / 32-channel block of the NEXT phase to stage
    auto stage = [&](int buf) {
        unsigned char *sa = smem + buf * PBUF;
        const int dy = st_r * p.dil - p.pad_y, dx = st_s * p.dil - p.pad_x;
        const long tapoff = ((long)dy * p.W + dx) * p.C * 2 + (long)st_c * 64;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const bool ok = rowok[i] && (unsigned)(oy[i] + dy) < (unsigned)p.H && (unsigned)(ox[i] + dx) < (unsigned)p.W;
            const uintptr_t src = reinterpret_cast<uintptr_t>(xb) + (uintptr_t)(xoff[i] + tapoff);
            const uintptr_t sel = ok ? src : reinterpret_cast<uintptr_t>(zp);  // a select, not a branch: ONE DMA per sub-tile
            lds_dma16(reinterpret_cast<const void *>(sel), sa + (wave * 2 + i) * SUB);
        }
        const long wk = ((long)(st_r * p.R + st_s) * p.C) * 2 + (long)st_c * 64;
#pragma unroll
        for (int j = 0; j < BL; j++) lds_dma16(wb + woff[j] + wk, sa + A_BYTES + wsub[j] * SUB);
        if (++st_c == kc2) {
            st_c = 0;
            if (++st_s == p.R) {
                st_s = 0;
                ++st_r;
            }
        }
    };

    float4_t acc[PT][CT];
#pragma unroll
    for (int i = 0; i < PT; i++)
#pragma unroll
        for (int j = 0; j < CT; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offset inside a sub-tile: row = lane & 15, k group = lane >> 4 (8 halves = 16 B), swizzled like the source
    const int fragoff = (lane & 15) * 64 + (((lane >> 4) * 16) ^ (((lane & 15) >> 3) << 5));

    // prologue: three phases on their way; phase 0 landed for everybody
    const int grp = wave >> 2;  // waves w and w + 4 share a SIMD: the two groups run half a phase apart (see the loop)
    for (int q = 0; q < 3 && q < np; q++) stage(q);
    if (np >= 3) wait_vmcnt<2 * NL>();
    else if (np == 2) wait_vmcnt<NL>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // Two sections per phase, a barrier after each: LOAD (issue the DMA of phase ph+3, read this phase's fragments from LDS,
    // retire the own DMA of phase ph+1) and MULTIPLY (32 MFMAs per wave).  Group 1 runs one barrier behind group 0, so on
    // every SIMD one wave multiplies while its partner loads: the matrix pipe does not wait for LDS / DMA issue.
    // Buffer (ph+3)&3 is the one phase ph-1 used; both groups finished reading it before the barrier that lets either of
    // them get here.  Phase ph+1's bytes are retired by EVERY wave one full phase before anyone reads them.
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int ph = 0; ph < np; ph++) {
        if (ph + 3 < np) stage((ph + 3) & (NBUF - 1));
        const unsigned char *sa = smem + (ph & (NBUF - 1)) * PBUF;
        const unsigned char *sb = sa + A_BYTES;
        half8_t wf[CT], xf[PT];
#pragma unroll
        for (int j = 0; j < CT; j++) wf[j] = *reinterpret_cast<const half8_t *>(sb + (wn * 4 + j) * SUB + fragoff);
#pragma unroll
        for (int i = 0; i < PT; i++) xf[i] = *reinterpret_cast<const half8_t *>(sa + (wm * PT + i) * SUB + fragoff);
        const int inflight = np - 1 - ph < 3 ? np - 1 - ph : 3;  // DMA groups of later phases outstanding now
        if (inflight == 3) wait_vmcnt<2 * NL>();
        else if (inflight == 2) wait_vmcnt<NL>();
        else if (inflight == 1) wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < PT; i++)
#pragma unroll
            for (int j = 0; j < CT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue from registers (16-byte stores: epilogue_store)
    epilogue_store<PT, CT>(acc, p, lane, n0 + wn * 64, [&](int i) -> long {
        const long m = m0 + wm * PM + i * 16 + (lane & 15);
        if (m >= p.M) return -1;
        if (!p.up_out) return m;
        // one output phase of the collapsed upsample convolution: (n, oy, ox) of the half-resolution grid -> (2 oy + py, 2 ox + px)
        const int n = (int)(m / HoWo), rem = (int)(m - (long)n * HoWo);
        const int oy_ = rem / p.Wo, ox_ = rem - oy_ * p.Wo;
        return ((long)n * (2 * p.Ho) + 2 * oy_ + p.py) * (2 * p.Wo) + 2 * ox_ + p.px;
    }, true);
}

// ------------------------------------------------------------------------------------------------ 3x3, halo tile in LDS
// The implicit-GEMM kernel above loads the pixel operand once PER TAP: nine times the same bytes, shifted.  A CU's load path
// (HBM / L2 -> LDS, ~12-20 B per clock) is what bounds it, not the matrix pipe.  For the 3x3 / pad = dilation = 1 layers --
// 90 % of the forward's convolution time -- this kernel loads each input pixel ONCE per 32-channel block: the workgroup's
// output tile is a TH x TW block of ONE image (512 pixels), its (TH+2) x (TW+2) input halo for the current channel block sits
// in LDS (double buffered: the next block's halo streams in during the nine taps of this one), and the nine taps are nine
// K-steps whose pixel fragments are read from the same halo at shifted pixel offsets.  Only the weights still stream per tap:
// BN = 128 output channels x 32 halves = 8 KiB per phase, next to ~5.6 KiB of halo, instead of 32 KiB per phase above.
// 8 waves as 4 (pixels) x 2 (channels), 128 x 64 outputs per wave (128 accumulator VGPRs), v_mfma_f32_16x16x32_f16 with the
// weight fragment as A and the pixel fragment as B as above; six-slot weight ring five slices ahead, counted vmcnt AND counted
// lgkmcnt, one raw barrier per phase, the phase body placed by hand (see the main loop), persistent workgroups (one per CU),
// the tile geometry a template parameter (LGTW = log2 of the tile width: 7, 6, 5, 4 for image widths >= 128, 64, 32, 16).
// Measured (tools/conv_ablate.py, 256 -> 256 @ 128 x 128, N = 128): 2.13 ms = 1.16 PFLOP/s; SQ_LDS_BANK_CONFLICT = 0.
constexpr int TP = 512;  // output pixels per workgroup

struct HaloParams {
    int TW, TH, lgTW;       // tile width (power of two, <= 128), height = 512 / TW
    int tiles_x, tiles;     // tiles per image row / per image
    int HWp, nhalo, npieces;  // halo row length TW + 2, halo pixels, 16-pixel DMA pieces (last one padded)
    int gimg, hrows;          // images per tile and halo rows per image: 1 / TH + 2, or -- maps lower than a tile (16 x 16) -- TH / H
                              // WHOLE images stacked in one tile, each with its own H + 2 halo rows
    int tps;                  // dilated form: tiles per row class (see D below)
};

// DIAGNOSTIC (MASK bit 1024): shader clock / 100 MHz reference clock stamps around the main loop of each workgroup's first tile
__device__ unsigned long long g_conv_clk[8 * 512];

// PH: -1 = the 3x3 convolution (nine taps per channel block).  0..3 = output phase (py, px) = (PH >> 1, PH & 1) of the COLLAPSED
// upsample convolution (pp_conv_up2_collapsed_f16): the input is the half-resolution tensor, the weights are (K, 2, 2, C) tap sums,
// only the FOUR taps of the 3x3 neighbourhood that phase sees are walked -- rows PH >> 1 + {0, 1}, columns PH & 1 + {0, 1} of the
// same halo -- and output pixel (y, x) of the tile lands at (2 y + py, 2 x + px) of a tensor twice the size.
// D: dilation (= padding) of the 3x3 convolution (models/layers_transposed.py:125-157, the backbone's DilatedConv 3, 3, 4, 4, 5, 5).
// Rows y, y +- D of an image never mix with the rows in between, so the image's rows are taken CLASS BY CLASS (y mod D): a tile is
// TH rows of ONE class -- real rows y0, y0 + D, y0 + 2 D, ... -- and its halo rows above and below are y0 - D and y0 + TH D:
// vertically the dilated convolution is the ordinary one on a permuted image.  Horizontally the halo row is TW + 2 D pixels long
// and the taps read it at column offsets 0, D, 2 D.  Tiles per class: ceil(ceil(H / D) / TH); rows past the image read zeros and
// store nothing (128 rows: 97 % / 100 % / 91 % of the tile rows are real for D = 3 / 4 / 5).
template <int BN, int MASK, int LGTW, int PH = -1, int D = 1>
__global__ __launch_bounds__(NTHREADS) void k_conv3x3_halo(const ConvParams p, const HaloParams hp) {
    constexpr int NT = PH < 0 ? 9 : 4;   // taps per channel block
    static_assert(D == 1 || PH < 0, "the collapsed upsample form has no dilation");
    // The tile geometry is a template parameter: every fragment address is then "lane register + immediate" (see R[][] below).
    constexpr int TW = 1 << LGTW, HWp = TW + 2 * D, TH = TP / TW;
    constexpr int dbg = MASK;  // ablation switches are COMPILE-TIME (a runtime switch costs a branch per guarded instruction); 0 in production
    constexpr int WN = BN / 64;          // 2
    constexpr int WM = 8 / WN;           // 4
    constexpr int PM = TP / WM;          // 128 pixels per wave
    constexpr int PT = PM / 16;          // 8
    constexpr int CT = 4;
    constexpr int WB = BN * 32 * 2;      // weight slice of one phase: 8 sub-tiles, one per wave
    constexpr int KEEP = 3;              // phases whose DMA may still be in flight when a phase ends (this one and 2 before)
    constexpr int NW = 6;                // weight ring
    constexpr int AHEAD = NW - 1;        // the slice of phase ph + AHEAD is issued while phase ph is multiplied, into the slot phase ph - 1 read
    constexpr int HP = 56;               // halo pieces per buffer: 7 per wave (pieces past the halo are zero-page reads)
#ifdef PP_HALO_NPW7   // A/B build (tools): seven pieces on every tile width
    constexpr int NPW = 7;
#else
    constexpr int NPW = (LGTW == 7 || D > 1) ? 7 : 6;   // pieces a wave really loads: the narrower tiles' halos (<= 45 pieces) need six
#endif
    static_assert(BN == 128, "one weight sub-tile per wave and phase");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int halo_bytes = HP * SUB;
    unsigned char *s_halo = smem;                               // [2][56 KiB]
    unsigned char *s_w = smem + 2 * halo_bytes;                 // [NW][WB]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // 1-D grid.  Workgroups b and b + 8 share an XCD (and its L2) under the round-robin placement: the channel tiles of ONE
    // pixel tile are 8 ids apart, so the second one finds the halo its sibling just fetched in L2 (speed only, never needed).
    // Each XCD (ids congruent mod 8) walks a CONTIGUOUS range of pixel tiles, all channel tiles of a pixel tile back to back:
    // the sibling channel tile and the vertically adjacent pixel tile (two shared halo rows) find their input in that L2.
    // PERSISTENT workgroups: the grid is one workgroup per CU (a multiple of 8); workgroup b takes the ids b, b + grid, ... --
    // all congruent mod 8, so it stays with "its" XCD's range.  The next tile's first DMA (halo of block 0, five weight slices)
    // is issued BEFORE the epilogue of the current tile: the output stores drain and the next operands arrive while the
    // epilogue computes, instead of store drain -> workgroup exit -> dispatch -> DMA latency in a row (a quarter of the time
    // of the non-persistent form).
    const int nct = (p.K + BN - 1) / BN;   // C_out = 64 mod 128: the upper half of the last tile has no channels (its weight
                                           // rows read as zeros through the buffer bounds, its epilogue is skipped)
    const int ptiles = (p.N / hp.gimg) * hp.tiles, per_xcd = (ptiles + 7) >> 3;
    const int total_ids = 8 * per_xcd * nct;
    const char *wb = reinterpret_cast<const char *>(p.w);
    int n_img = 0, ty0 = 0, tx0 = 0, n0 = 0, t_idx = 0;   // the tile being STAGED (wave-uniform); t_idx: its index inside the image
    const char *xb = nullptr;
    // ---- halo DMA sources.  A piece is 16 halo pixels x 64 B, lane-linear in LDS; the swizzle is applied on the source side.
    // Halo swizzle: bit 5 ^= bit 8 of the byte address inside the halo buffer, i.e. the two 32-B halves of a pixel swap on
    // every other group of FOUR pixels.  With it a ds_read_b128 of 16 consecutive pixels x 4 k-groups is bank-conflict free
    // for EVERY start pixel (taps shift the start by 0 / 1 / 2 pixels and by the halo row length); the weight image's
    // swizzle (bit 9) would make 14 of 16 start offsets 2-way conflicted.
    // Wave w loads pieces w, w + 8, ..., w + 48 of every channel block (see the DMA schedule).
    int hsrc[7];   // byte offset inside the image for channel block 0, or -1: zero page (outside the image / past the halo)
    // A halo takes 39 .. 52 pieces: pieces 4, 5, 6 of a wave (32 + wave, 40 + wave, 48 + wave) exist for some waves only -- e.g.
    // 49 pieces on the 128-wide tiles: a seventh for wave 0 alone; 42 on the 64-wide: a sixth for waves 0 and 1.  A wave skips the
    // instructions of the pieces it does not own -- a load of nothing still costs its issue slot on the CU's one load path -- and
    // counts that many loads less in the windows that hold them (`missing` below).
    const bool p4 = __builtin_amdgcn_readfirstlane(32 + wave < hp.npieces);
    const bool p5 = __builtin_amdgcn_readfirstlane(40 + wave < hp.npieces);
    const bool p6 = __builtin_amdgcn_readfirstlane(48 + wave < hp.npieces);
    unsigned woff = 0; // weight sub-tile of this wave: 16 output channels x 32 halves per phase (lane part of the offset)
    int bx_lo = 0, bx_hi = 0, bw_lo = 0, bw_hi = 0, nx = 0, bw_n = 0;   // buffer bases (this image / this wave's 16 weight rows) and the image's bytes
    auto rsrc_of = [&](int lo, int hi, int nbytes) {
        void *base = reinterpret_cast<void *>(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
        return __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
    };
    auto decode = [&](int id) -> bool {   // id -> tile; false: padding of an XCD's range
        const int xcd = id & 7, q = id >> 3;
        const int ptile = xcd * per_xcd + q / nct, ctile = q - (q / nct) * nct;
        if (q / nct >= per_xcd || ptile >= ptiles) return false;
        n_img = (ptile / hp.tiles) * hp.gimg;   // the tile's (first) image
        const int tile = ptile - (ptile / hp.tiles) * hp.tiles;
        t_idx = tile;
        if (D == 1) ty0 = (tile / hp.tiles_x) * TH;
        else {   // row class c = (tile row) / tps, tile t of that class: real rows c + (t TH + 0 .. TH - 1) D
            const int trow_ = tile / hp.tiles_x, cls = trow_ / hp.tps;
            ty0 = cls + (trow_ - cls * hp.tps) * TH * D;
        }
        tx0 = (tile - (tile / hp.tiles_x) * hp.tiles_x) * TW;
        n0 = ctile * BN;
        const int sw_ = p.up ? p.W >> 1 : p.W;                                       // source row length in pixels
        const int ldx = p.ldx ? p.ldx : p.C;                                         // elements between consecutive input pixels
        xb = reinterpret_cast<const char *>(p.x) + (long)n_img * (p.up ? p.H >> 1 : p.H) * sw_ * ldx * 2;   // this image
        int ln = lane;
        asm volatile("" : "+v"(ln));   // recomputed per tile: hoisted, the per-piece halo coordinates would sit in 20 registers
#pragma unroll
        for (int t = 0; t < NPW; t++) {
            const int piece = t * 8 + wave;
            const int phys = piece * SUB + ln * 16;
            const int logical = phys ^ (((phys >> 8) & 1) << 5);
            const int hpix = logical >> 6, chunk = (logical >> 4) & 3;
            const int hy = hpix / HWp, hx = hpix - hy * HWp;
            const int gi = D == 1 ? hy / hp.hrows : 0;                 // image of the tile this halo row belongs to (0 unless stacked)
            const int iy = D == 1 ? ty0 - 1 + (hy - gi * hp.hrows) : ty0 + (hy - 1) * D, ix = tx0 - D + hx;
            const bool ok = hpix < hp.nhalo && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hsrc[t] = ok ? (((gi * (p.H >> p.up) + (iy >> p.up)) * sw_ + (ix >> p.up)) * ldx * 2 + chunk * 16)
                         : (int)0x80000000;   // outside num_records: zeros
        }
        const int b = ln * 16;
        const int bs = b ^ (((b >> 9) & 1) << 5);
        woff = (unsigned)((bs >> 6) * (NT * p.C * 2) + (bs & 63));   // lane part: row of the sub-tile, byte inside the 32-half k slice
        // buffer resources (scalar): this image (bounds = the image: border pixels read zeros), this wave's 16 weight rows
        // (kept as explicitly wave-uniform dwords: a resource the compiler cannot prove uniform costs a waterfall loop per load)
        const unsigned long long ax = reinterpret_cast<unsigned long long>(xb);
        const unsigned long long aw = reinterpret_cast<unsigned long long>(wb) + (unsigned long long)((long)(n0 + wave * 16) * ((long)NT * p.C * 2));
        bx_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)ax), bx_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(ax >> 32));
        bw_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)aw), bw_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(aw >> 32));
        nx = __builtin_amdgcn_readfirstlane(hp.gimg * (p.up ? p.H >> 1 : p.H) * sw_ * ldx * 2);
        bw_n = __builtin_amdgcn_readfirstlane(n0 + wave * 16 < p.K ? 16 * NT * p.C * 2 : 0);   // bytes of this wave's 16 weight rows
        return true;
    };
    auto next_tile = [&](int id) -> int {   // first valid id at or after `id` on this workgroup's stride, or total_ids
        while (id < total_ids && !decode(id)) id += gridDim.x;
        return id;
    };

    const int ncb = p.C / 32;
    const int np = ncb * NT;
    int st_cb = 0, st_tap = 0;   // (channel block, tap) of the NEXT weight slice to stage
    int st_slot = 0;
    int slot_cur = 0;            // ring slot of the phase being multiplied
    auto stage_w = [&]() {
        lds_dma16_buf(rsrc_of(bw_lo, bw_hi, bw_n), woff, (unsigned)((st_tap * p.C + st_cb * 32) * 2), s_w + st_slot * WB + wave * SUB);
        if (++st_slot == NW) st_slot = 0;
        if (++st_tap == NT) {
            st_tap = 0;
            ++st_cb;
        }
    };
    auto stage_halo = [&](int t, int cb, int buf) {   // piece t * 8 + wave (t compile-time 0..6) of channel block cb
        // 32-bit offsets inside the image (< 2^31 bytes: checked by the launcher); the channel block is the scalar offset
        lds_dma16_buf(rsrc_of(bx_lo, bx_hi, nx), (unsigned)hsrc[t], (unsigned)(cb * 64), s_halo + buf * halo_bytes + (t * 8 + wave) * SUB);
    };

    // fragment addresses.  Weights: sub-tile image as above.  Pixels: lane reads 16 B (k group g) of halo pixel
    // (qy + r) * HWp + qx + s for its output pixel q; pixel tile i starts at tile pixel wm * 128 + 16 i (a multiple of 16 <= TW).
    // With L = hb0 + C the unswizzled byte offset (hb0 the lane part, C = 64 x the pixel offset of (tap, pixel tile): compile-time),
    // the swizzled address L ^ (bit8(L) << 5) equals  (swz(hb0 + 64 k) ^ (f << 5)) + Chi  with k = (C / 64) & 3, Chi = C & ~255,
    // f = bit 8 of Chi -- adding a multiple of 256 neither carries into bit 8 from below nor touches bit 5.  Eight lane registers
    // R[k][f] (set up per tile, moved to the other halo buffer per channel block) and the 16-bit immediate offset Chi: a
    // fragment read costs NO address arithmetic (it was four VALU instructions per read, 32 per wave and phase).
    const int wfrag = (lane & 15) * 64 + (((lane >> 4) * 16) ^ (((lane & 15) >> 3) << 5));
    // (stacked images: the wave's rows lie in image (rows before) / H of the tile, whose halo starts two rows per image further down)
    const int rows_before = (wm * PM) >> LGTW;
    const int hb0 = ((rows_before + (hp.gimg > 1 ? 2 * (rows_before / p.H) : 0)) * HWp + ((wm * PM) & (TW - 1)) + (lane & 15)) * 64 +
                    (lane >> 4) * 16;
    // LDS byte addresses (32-bit) of the two halo buffers and the weight ring, for the hand-placed ds_read_b128 below
    const unsigned lds_halo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_halo;
    const unsigned lds_w = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_w + wn * 4 * SUB;
    unsigned R[4][2];
    auto set_R = [&]() {   // for halo buffer 0
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int M = hb0 + k * 64;
            const unsigned a = lds_halo + (unsigned)(M ^ ((M >> 3) & 32));
            R[k][0] = a;
            R[k][1] = a ^ 32u;
        }
    };
    auto flip_R = [&](int to_buf) {   // to the other halo buffer
        const unsigned d = to_buf ? (unsigned)halo_bytes : (unsigned)-halo_bytes;
#pragma unroll
        for (int k = 0; k < 4; k++) R[k][0] += d, R[k][1] += d;
    };
    auto xread = [&](auto TAP, auto I, half8_t &dst) {   // pixel fragment of (tap, pixel tile i) from the halo buffer R points to
        constexpr int tap = decltype(TAP)::value, i = decltype(I)::value;
        constexpr int trow = PH < 0 ? tap / 3 : (PH >> 1) + (tap >> 1), tcol = PH < 0 ? D * (tap % 3) : (PH & 1) + (tap & 1);
        constexpr int Cpix = trow * HWp + tcol + ((i * 16) >> LGTW) * HWp + ((i * 16) & (TW - 1));
        constexpr int C = Cpix * 64, k = Cpix & 3, Chi = C & ~255, f = (Chi >> 8) & 1;
        static_assert(Chi < 65536, "16-bit DS offset");
        lds_read16<Chi>(dst, R[k][f]);
    };

    // first DMA of a tile: halo of channel block 0 (7 pieces per wave) into buffer 0 and the weight slices of phases 0..AHEAD-1
    auto stage_first = [&]() {
        st_cb = 0, st_tap = 0, st_slot = 0;
#pragma unroll
        for (int t = 0; t < NPW; t++)
            if (t < 4 || (t == 4 && p4) || (t == 5 && p5) || (t == 6 && p6)) stage_halo(t, 0, 0);
        for (int q = 0; q < AHEAD && q < np; q++) stage_w();
    };
    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;   // DIAGNOSTIC (MASK bit 1024): cycles per section, summed over tiles
    auto stamp = [&](int k) {
        if (dbg & 1024) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) acc_t[k] += t - t_prev;
            t_prev = t;
        }
    };
    int id = next_tile(blockIdx.x);
    if (id >= total_ids) return;
    // Start stagger: workgroups of equal work started together reach their epilogues together -- the whole chip stores (HBM
    // write-bound, matrix pipes idle), then the whole chip multiplies (HBM idle).  Spreading the starts over one tile's time
    // lets one CU's store burst run under its neighbours' MFMAs.
    for (int z = (int)(blockIdx.x >> 3) * p.stagger / (int)((gridDim.x + 7) >> 3); z > 0; z--) __builtin_amdgcn_s_sleep(127);
    stage_first();
  while (true) {
    const int c_img = n_img, c_ty0 = ty0, c_tx0 = tx0, c_n0 = n0, c_tile = t_idx;   // the tile being COMPUTED
    float4_t acc[PT][CT];
#pragma unroll
    for (int i = 0; i < PT; i++)
#pragma unroll
        for (int j = 0; j < CT; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    stamp(-1);
    wait_vmcnt<0>();     // first DMA of this tile landed (and the previous tile's stores are out)
    __builtin_amdgcn_s_barrier();
    stamp(0);            // 0: wait for first DMA + stores, barrier
    slot_cur = 0;
    half8_t wf[CT], xf[PT];
    set_R();
    lds_read16<0>(wf[0], lds_w + wfrag);
    lds_read16<SUB>(wf[1], lds_w + wfrag);
    using T0 = std::integral_constant<int, 0>;
    xread(T0{}, std::integral_constant<int, 0>{}, xf[0]);
    xread(T0{}, std::integral_constant<int, 1>{}, xf[1]);
    xread(T0{}, std::integral_constant<int, 2>{}, xf[2]);
    xread(T0{}, std::integral_constant<int, 3>{}, xf[3]);
    xread(T0{}, std::integral_constant<int, 4>{}, xf[4]);
    xread(T0{}, std::integral_constant<int, 5>{}, xf[5]);
    xread(T0{}, std::integral_constant<int, 6>{}, xf[6]);
    xread(T0{}, std::integral_constant<int, 7>{}, xf[7]);
    wait_lgkmcnt<0>();

    // ---- main loop: channel blocks x 9 taps (unrolled: the DMA count of a tap is a compile-time constant).
    // The MFMAs and fragment reads of a phase are placed BY HAND (asm volatile keeps their order; "+v" keeps every accumulator
    // in its own four registers -- the compiler's allocation of the builtin form rotated the accumulators through ~50 spare
    // registers, spilled the fragment address and, because a scratch reload counts in vmcnt like the LDS-DMA, put an
    // s_waitcnt vmcnt(0) -- i.e. "all DMA landed" -- three times into every phase).
    // In-place fragment refill, every register reloaded >= 16 MFMAs (256 cycles) before its next use:
    //   half A: 8 pixel tiles x channel tiles 0, 1;  at its start weight fragments 2, 3 of THIS phase are read (their registers
    //           became free when the previous phase ended);
    //   half B: 8 pixel tiles x channel tiles 2, 3;  at its start weight fragments 0, 1 of the NEXT phase are read, and after
    //           the two MFMAs of pixel tile i its fragment of the next phase (next tap: the same halo, shifted).
    // LDS returns data in order, so lgkmcnt is counted like vmcnt: before pixel tile i of half A the reads issued after ITS
    // fragment are tiles i+1..7 of the previous phase and this phase's two weight reads: lgkmcnt(9 - i).
    // DMA schedule of a wave, per tap: one weight sub-tile (slice ph + AHEAD, into the ring slot phase ph - 1 read from), plus,
    // while a next channel block exists, its halo pieces: 2, 2, 1, 1 (, 1) at taps 0..3 (4) -- every piece is KEEP phases old when tap
    // 7 ends.  After the MFMAs the wave waits until only the DMA of the last KEEP phases is in flight (counted vmcnt; the
    // count is the schedule's, a compile-time sum), then the barrier publishes what landed: the weight slice of phase ph + 2,
    // whose fragments 0, 1 are read in half B of the next phase.  No dummy DMA anywhere.
    int ph = 0;
    const int grp = wave >> 2;   // waves w and w + 4 share a SIMD
    unsigned long long rt0 = 0;
    if (dbg & 1024) rt0 = __builtin_amdgcn_s_memrealtime();
    stamp(1);            // 1: first fragment reads
    auto phase = [&](auto TAP, auto HALO, int cb) {
        constexpr int tap = decltype(TAP)::value;
        constexpr bool halo = decltype(HALO)::value;     // a next channel block exists: its halo streams in
        constexpr int ntap = tap == NT - 1 ? 0 : tap + 1;
        // halo pieces issued per tap.  A piece is waited for KEEP = 3 taps after its issue and must have landed when the block ends:
        // nine taps spread them 2-2-1-1(-1) over taps 0..3 (4), four taps issue all NPW at tap 0.
        constexpr auto halo_at = [](int t) { return t < 0 ? 0 : (NT == 9 ? (t <= 1 ? 2 : (t <= NPW - 3 ? 1 : 0)) : (t == 0 ? NPW : 0)); };
        // The NEW halo buffer is first read by the fragment refill in half B of the block's LAST tap (after flip_R), so its pieces
        // must have landed when the tap before that ends.  Nine taps: the last piece is issued at tap 4 and is KEEP phases old
        // when tap 7 ends.  Four taps: every piece is issued at tap 0 and must be down when tap 2 ends -- only the three weight
        // slices issued after them may still be in flight there (counting the pieces of "the last three taps" at tap 2 let the
        // refill read a halo that was still arriving: wrong pixels under load, found by tools/collapsed_repeat.py).
        constexpr int in_flight = KEEP + (!halo ? 0 : NT == 9 ? halo_at(tap) + halo_at(tap - 1) + halo_at(tap - 2) : (tap <= 1 ? NPW : 0));
        static_assert(KEEP == 3, "in_flight sums the halo pieces of three taps");
        // The two waves of a SIMD (w and w + 4) issue their DMA at different times: a piece blocks its wave for 60-100 cycles, and
        // with both waves there at once right after the barrier the matrix pipe idles; group 1 issues between the halves.
        auto issue_dma = [&]() {
            if (!(dbg & 1)) {
                if (halo) {
                    if (NT == 4) {
                        if (tap == 0) {
                            stage_halo(0, cb + 1, (cb + 1) & 1);
                            stage_halo(1, cb + 1, (cb + 1) & 1);
                            stage_halo(2, cb + 1, (cb + 1) & 1);
                            stage_halo(3, cb + 1, (cb + 1) & 1);
                            if (p4) stage_halo(4, cb + 1, (cb + 1) & 1);
                            if (p5) stage_halo(5, cb + 1, (cb + 1) & 1);
                            if (NPW == 7 && p6) stage_halo(6, cb + 1, (cb + 1) & 1);
                        }
                    } else if (tap <= 1) {
                        stage_halo(2 * tap, cb + 1, (cb + 1) & 1);
                        stage_halo(2 * tap + 1, cb + 1, (cb + 1) & 1);
                    } else if (tap <= NPW - 3) {
                        if (tap == 2 ? p4 : (tap == 3 ? p5 : p6)) stage_halo(tap + 2, cb + 1, (cb + 1) & 1);
                    }
                }
                if (ph + AHEAD < np) stage_w();
            }
        };
        if (grp == 0 || (dbg & 128)) issue_dma();
        const unsigned sw_cur = lds_w + slot_cur * WB + wfrag;
        if (++slot_cur == NW) slot_cur = 0;
        const unsigned sw_nxt = lds_w + slot_cur * WB + wfrag;
        __builtin_amdgcn_s_setprio(1);
        if (!(dbg & 4)) {
            lds_read16<2 * SUB>(wf[2], sw_cur);
            lds_read16<3 * SUB>(wf[3], sw_cur);
        }
        auto tile_a = [&](auto I) {
            constexpr int i = decltype(I)::value;
            wait_lgkmcnt<9 - i>();
            if (!(dbg & 2)) {
                mfma_acc(acc[i][0], wf[0], xf[i]);
                mfma_acc(acc[i][1], wf[1], xf[i]);
            }
        };
        tile_a(std::integral_constant<int, 0>{});
        tile_a(std::integral_constant<int, 1>{});
        tile_a(std::integral_constant<int, 2>{});
        tile_a(std::integral_constant<int, 3>{});
        tile_a(std::integral_constant<int, 4>{});
        tile_a(std::integral_constant<int, 5>{});
        tile_a(std::integral_constant<int, 6>{});
        tile_a(std::integral_constant<int, 7>{});
        static_assert(PT == 8, "eight pixel tiles per wave");
        constexpr bool refill = !(dbg & 4) && !(tap == NT - 1 && !halo);   // the last phase of all has no successor (a read whose
                                                                      // result nobody waits for may land in a re-used register)
        if (grp != 0 && !(dbg & 128)) issue_dma();
        if (refill) {
            lds_read16<0>(wf[0], sw_nxt);
            lds_read16<SUB>(wf[1], sw_nxt);
        }
        wait_lgkmcnt<refill ? 2 : 0>();
        if (refill && tap == NT - 1) flip_R((cb + 1) & 1);   // the next phase is tap 0 of the next channel block: the other halo buffer
        auto tile_b = [&](auto I) {
            constexpr int i = decltype(I)::value;
            if (!(dbg & 2)) {
                mfma_acc(acc[i][2], wf[2], xf[i]);
                mfma_acc(acc[i][3], wf[3], xf[i]);
            }
            if (refill) xread(std::integral_constant<int, ntap>{}, I, xf[i]);
        };
        tile_b(std::integral_constant<int, 0>{});
        tile_b(std::integral_constant<int, 1>{});
        tile_b(std::integral_constant<int, 2>{});
        tile_b(std::integral_constant<int, 3>{});
        tile_b(std::integral_constant<int, 4>{});
        tile_b(std::integral_constant<int, 5>{});
        tile_b(std::integral_constant<int, 6>{});
        tile_b(std::integral_constant<int, 7>{});
        __builtin_amdgcn_s_setprio(0);
        // pieces this wave did NOT issue inside the window of this tap's wait (piece 4 / 5 / 6 goes out at tap 2 / 3 / 4 of nine and
        // stays in the window for three taps; all go out at tap 0 of four and stay for taps 0 and 1): wave-uniform, 0 .. 3
        constexpr bool w4 = halo && (NT == 9 ? (tap >= 2 && tap <= 4) : tap <= 1);
        constexpr bool w5 = halo && (NT == 9 ? (tap >= 3 && tap <= 5) : tap <= 1);
        constexpr bool w6 = halo && NPW == 7 && (NT == 9 ? (tap >= 4 && tap <= 6) : tap <= 1);
        const int missing = (w4 && !p4) + (w5 && !p5) + (w6 && !p6);
        static_assert(in_flight >= 3, "the window always holds the three weight slices");
        if (ph + AHEAD >= np) wait_vmcnt<0>();     // the last phases issue nothing: nothing to wait for
        else if (missing == 0) wait_vmcnt<in_flight>();
        else if (missing == 1) wait_vmcnt<(in_flight > 1 ? in_flight - 1 : 0)>();
        else if (missing == 2) wait_vmcnt<(in_flight > 2 ? in_flight - 2 : 0)>();
        else wait_vmcnt<(in_flight > 3 ? in_flight - 3 : 0)>();
        __builtin_amdgcn_s_barrier();
        ++ph;
    };
    auto block = [&](auto HALO, int cb) {
        phase(std::integral_constant<int, 0>{}, HALO, cb);
        phase(std::integral_constant<int, 1>{}, HALO, cb);
        phase(std::integral_constant<int, 2>{}, HALO, cb);
        phase(std::integral_constant<int, 3>{}, HALO, cb);
        if constexpr (NT == 9) {
            phase(std::integral_constant<int, 4>{}, HALO, cb);
            phase(std::integral_constant<int, 5>{}, HALO, cb);
            phase(std::integral_constant<int, 6>{}, HALO, cb);
            phase(std::integral_constant<int, 7>{}, HALO, cb);
            phase(std::integral_constant<int, 8>{}, HALO, cb);
        }
    };
    for (int cb = 0; cb + 1 < ncb; cb++) block(std::true_type{}, cb);
    block(std::false_type{}, ncb - 1);
    // the matrix pipe may still be writing the last accumulators: the compiler does not see MFMAs in the asm statements
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    stamp(2);            // 2: main loop
    if (dbg & 1024) acc_t[5] += __builtin_amdgcn_s_memrealtime() - rt0;   // 5: main loop in 100 MHz ticks

    // ---- next tile's first DMA (every LDS read of this tile is behind the last barrier), then this tile's epilogue from
    // registers (16-byte stores: epilogue_store)
    id = next_tile(id + gridDim.x);
    const bool has_next = id < total_ids;
    if (has_next) stage_first();
    stamp(3);            // 3: next tile's decode + first DMA issue
    if (dbg & 64)
        epilogue_fullline_probe<PT, CT>(acc, p, lane, c_n0 + wn * 64, [&](int qq) -> long {
            const int q = wm * PM + qq;
            const int qy = q >> LGTW, qx = q & (TW - 1);
            return ((long)c_img * p.H + c_ty0 + qy) * p.W + c_tx0 + qx;
        });
    else
    epilogue_store<PT, CT>(acc, p, lane, c_n0 + wn * 64, [&](int i) -> long {
        const int q = wm * PM + i * 16 + (lane & 15);
        const int qy = q >> LGTW, qx = q & (TW - 1);
        if (PH >= 0)   // collapsed upsample convolution: this phase's place in the tensor of twice the size (rows counted globally)
            return (2 * ((long)c_img * p.H + c_ty0 + qy) + (PH >> 1)) * (2 * p.W) + 2 * (c_tx0 + qx) + (PH & 1);
        if (D > 1) {   // tile row qy is real row y0 + qy D of its class; the class may end inside the tile
            const int ry = c_ty0 + qy * D;
            return ry < p.H ? ((long)c_img * p.H + ry) * p.W + c_tx0 + qx : -1;
        }
        return ((long)c_img * p.H + c_ty0 + qy) * p.W + c_tx0 + qx;
    }, !(dbg & 8), p.csum ? p.csum + (((long)c_img * hp.tiles + c_tile) * WM + wm) * p.K : nullptr);
    stamp(4);            // 4: epilogue (issue)
    if (!has_next) break;
  }
    if ((dbg & 1024) && threadIdx.x == 0 && blockIdx.x < 512)
        for (int k = 0; k < 6; k++) g_conv_clk[blockIdx.x * 8 + k] = acc_t[k];
}

// Per-DEVICE lazily initialised state (a process may hold contexts on several GPUs): the zero page the implicit-GEMM kernel
// reads out-of-image taps from, the CU count of the persistent grid, and which kernel instances have had their dynamic-LDS
// attribute raised on that device.  Indexed by hipGetDevice(); initialisation allocates, so it refuses to run inside a stream
// capture (PP_ERR_UNSUPPORTED: call the entry point once eagerly first, as every warm-up does).
constexpr int kMaxDevices = 32, kMaxInst = 192;
struct DevState {
    void *zero = nullptr;
    int ncu = 0;
    bool attr[kMaxInst] = {};
};
DevState g_dev[kMaxDevices];
std::atomic<int> g_inst_count{0};   // instance ids handed out to the launch templates (one per instantiation)

DevState *dev_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return &g_dev[dev];
}
bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
// raises the dynamic-LDS limit of `fn` once per device; inst: the instantiation's slot in DevState::attr
int ensure_attr(const void *fn, int lds, int inst, hipStream_t st) {
    DevState *d = dev_state();
    if (!d || inst < 0 || inst >= kMaxInst) return PP_ERR_HIP;
    if (d->attr[inst]) return PP_OK;
    if (capturing(st)) return PP_ERR_UNSUPPORTED;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return PP_ERR_HIP;
    d->attr[inst] = true;
    return PP_OK;
}

// geometry of the halo kernel for an image size, or false when the shape is not taken (the implicit-GEMM kernel runs it)

bool halo_geometry(const ConvParams &p, HaloParams &g) {
    if (p.R != 3 || p.pad != p.dil || p.dil < 1 || p.C % 32 || p.K % 64) return false;   // K = 64 mod 128: the last channel tile is half empty
    if (p.dil != 1 && (p.dil < 3 || p.dil > 5 || p.up || p.csum)) return false;          // dilated instances: 3, 4, 5 (the backbone's)
    if ((long)p.H * p.W * (p.ldx ? p.ldx : p.C) * 2 >= (1L << 31)) return false;   // 32-bit byte offsets inside one image
    int tw = 128;   // the widest power-of-two tile (<= 128) that divides the image width: 128, 64 (e.g. W = 192), 32 (96), 16 (48)
    while (tw >= 16 && p.W % tw) tw >>= 1;
    if (tw < 16) return false;
    const int th = TP / tw;
    g.TW = tw;
    g.TH = th;
    g.lgTW = 0;
    while ((1 << g.lgTW) < tw) g.lgTW++;
    g.HWp = tw + 2 * p.dil;
    g.tps = 0;
    if (p.dil > 1) {   // rows class by class (see k_conv3x3_halo): D classes of ceil(ceil(H / D) / th) tiles each
        if (tw < 32) return false;   // instances exist for the 128-, 64- and 32-wide tiles
        g.gimg = 1;
        g.hrows = th + 2;
        g.tiles_x = p.W / tw;
        g.tps = ((p.H + p.dil - 1) / p.dil + th - 1) / th;
        g.tiles = g.tiles_x * p.dil * g.tps;
        g.nhalo = (th + 2) * (tw + 2 * p.dil);
    } else if (p.H % th == 0) {
        g.gimg = 1;
        g.hrows = th + 2;
        g.tiles_x = p.W / tw;
        g.tiles = g.tiles_x * (p.H / th);
        g.nhalo = (th + 2) * (tw + 2);
    } else {
        // a map lower than the tile (16 x 16: tile 16 wide x 32 rows): th / H whole images per tile, stacked, every wave's 128
        // pixels inside one of them
        if (tw != p.W || th % p.H || p.H % (128 / tw) || p.N % (th / p.H) || p.csum) return false;
        g.gimg = th / p.H;
        g.hrows = p.H + 2;
        g.tiles_x = 1;
        g.tiles = 1;
        g.nhalo = g.gimg * (p.H + 2) * (tw + 2);
    }
    g.npieces = (g.nhalo + 15) / 16;
    // 7 pieces per wave and channel block (six for the narrower tiles: NPW); the first four of every wave always exist
    return g.npieces >= 32 && g.npieces <= (g.lgTW == 7 || p.dil > 1 ? 56 : 48);
}



template <int MASK, int LGTW, int PH = -1, int D = 1>
int launch_halo_inst(const ConvParams &p, const HaloParams &g, hipStream_t st) {
    const int lds = 2 * 56 * SUB + 6 * (128 * 32 * 2);   // two halo buffers of 56 pieces, weight ring of 6 slices
    static const int inst = g_inst_count.fetch_add(1);
    if (const int rc = ensure_attr(reinterpret_cast<const void *>(&k_conv3x3_halo<128, MASK, LGTW, PH, D>), lds, inst, st)) return rc;
    const unsigned ptiles = (unsigned)((p.N / g.gimg) * g.tiles);
    const unsigned ids = ((ptiles + 7) / 8) * 8 * (unsigned)((p.K + 127) / 128);   // 8 XCD ranges x ceil(ptiles / 8) x channel tiles
    DevState *ds = dev_state();   // persistent grid: one workgroup per CU (160 KiB of LDS each), a multiple of 8
    if (!ds) return PP_ERR_HIP;
    if (ds->ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return PP_ERR_HIP;
        ds->ncu = n >= 8 ? (n / 8) * 8 : 8;
    }
    const int ncu = ds->ncu;
    static const bool persist = [] {
        const char *e = getenv("POSEPAF_CONV_PERSIST");   // diagnostics: 0 = one tile per workgroup
        return !(e && e[0] == '0');
    }();
    const dim3 grid(persist && ids > (unsigned)ncu ? (unsigned)ncu : ids);
    ConvParams q = p;
    if (q.stagger < 0) {   // default: spread the starts over half a tile's time (np phases of ~1500 cycles), less when a workgroup has few tiles
        const float rounds = (float)ids / (float)grid.x;
        const float f = 0.5f * (rounds < 32.f ? rounds / 32.f : 1.f);
        q.stagger = grid.x < ids ? (int)(f * (float)(p.C / 32 * (PH < 0 ? 9 : 4)) * 1500.f / 8128.f + 0.5f) : 0;
    }
    hipLaunchKernelGGL((k_conv3x3_halo<128, MASK, LGTW, PH, D>), grid, dim3(NTHREADS), lds, st, q, g);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

template <int D>
int launch_halo_dilated(const ConvParams &p, const HaloParams &g, hipStream_t st) {
    switch (g.lgTW) {
        case 7: return launch_halo_inst<0, 7, -1, D>(p, g, st);
        case 6: return launch_halo_inst<0, 6, -1, D>(p, g, st);
        case 5: return launch_halo_inst<0, 5, -1, D>(p, g, st);
        default: return PP_ERR_UNSUPPORTED;
    }
}

template <int MASK>
int launch_halo_mask(const ConvParams &p, const HaloParams &g, hipStream_t st) {
    if (p.dil != 1) {
        if (MASK != 0) return PP_ERR_UNSUPPORTED;
        return p.dil == 3 ? launch_halo_dilated<3>(p, g, st) : (p.dil == 4 ? launch_halo_dilated<4>(p, g, st) : launch_halo_dilated<5>(p, g, st));
    }
    if (g.lgTW == 7) return launch_halo_inst<MASK, 7>(p, g, st);
    if (MASK == 0) {   // the ablated instances exist for the 128-wide tile only
        if (g.lgTW == 6) return launch_halo_inst<0, 6>(p, g, st);
        if (g.lgTW == 5) return launch_halo_inst<0, 5>(p, g, st);
        if (g.lgTW == 4) return launch_halo_inst<0, 4>(p, g, st);
    }
    return PP_ERR_UNSUPPORTED;
}

template <int PH>
int launch_halo_phase(const ConvParams &p, const HaloParams &g, hipStream_t st) {
    switch (g.lgTW) {
        case 7: return launch_halo_inst<0, 7, PH>(p, g, st);
        case 6: return launch_halo_inst<0, 6, PH>(p, g, st);
        case 5: return launch_halo_inst<0, 5, PH>(p, g, st);
        case 4: return launch_halo_inst<0, 4, PH>(p, g, st);
        default: return PP_ERR_UNSUPPORTED;
    }
}

int launch_halo(const ConvParams &p, const HaloParams &g, hipStream_t st) {
    // POSEPAF_CONV_DBG (diagnostics): compile-time ablated instances; 0 = the product.  "No MFMA" exists only together with "no
    // fragment reads" (6, 7): a hand-placed ds_read whose result nothing consumes may land in a register the compiler has re-used.
    switch (p.dbg) {
        case 0: return launch_halo_mask<0>(p, g, st);
#ifdef PP_CONV_DIAG
        case 1: return launch_halo_mask<1>(p, g, st);     // no DMA inside the loop
        case 4: return launch_halo_mask<4>(p, g, st);     // no fragment reads
        case 6: return launch_halo_mask<6>(p, g, st);     // DMA + barriers only
        case 5: return launch_halo_mask<5>(p, g, st);     // MFMA + barriers only
        case 7: return launch_halo_mask<7>(p, g, st);     // barriers only
        case 15: return launch_halo_mask<15>(p, g, st);   // ... and no epilogue
        case 64: return launch_halo_mask<64>(p, g, st);   // full-line store pattern (wrong placement: timing only)
        case 71: return launch_halo_mask<71>(p, g, st);   // ... with barriers only
        case 1152: return launch_halo_mask<1152>(p, g, st);   // stamps, both waves of a SIMD issue DMA at the phase start
        case 1024: return launch_halo_mask<1024>(p, g, st);   // in-kernel clock stamps (pp_conv_debug_clock)
        case 1028: return launch_halo_mask<1028>(p, g, st);
        case 1029: return launch_halo_mask<1029>(p, g, st);
#endif
        default: return PP_ERR_BAD_ARG;
    }
}

template <int BN>
int launch(const ConvParams &p, hipStream_t st) {
    constexpr int lds = 4 * (BM * 32 * 2 + BN * 32 * 2);  // four phase buffers
    static const int inst = g_inst_count.fetch_add(1);
    if (const int rc = ensure_attr(reinterpret_cast<const void *>(&k_conv_igemm<BN>), lds, inst, st)) return rc;
    const dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)(p.K / BN));
    hipLaunchKernelGGL(k_conv_igemm<BN>, grid, dim3(NTHREADS), lds, st, p);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}


// ------------------------------------------------------------------------------------------------ 1x1, streaming
// The 1x1 convolutions of the IMHN (bottleneck c1 / c3 of models/layers_transposed.py:12-48, the merge convolutions and heads of
// models/posenet.py:60-118) are HBM-bound: 50 flop per byte against the chip's ~300.  What costs is bytes and passes, so this
// kernel is a stream with the neighbouring element-wise passes folded in:
//   y = act(conv1x1(x * scale[n]) + bias (+ extra))   and optionally   y2 = y + extra2
//   * `scale` (n, K): the SE block's per-sample channel gains (models/layers_transposed.py:289-310) applied to the INPUT fragment
//     in registers -- binary16 products, i.e. exactly the tensor a separate x * s pass would have written; that pass disappears;
//   * y2: `x + cache` next to `cache` (models/posenet.py:116-118) from one pass;
//   * ldy: y may be a channel slice of a wider tensor.
// Structure: persistent workgroups of 8 waves, the weights of the workgroup's output-channel range resident in LDS (sub-tiles of
// 16 channels x 32 k, swizzled like k_conv_igemm's), NO barrier after that: every wave walks its own groups of 32 pixels --
// the pixel fragments of the MFMA come straight from global memory into registers (lane = pixel x 8-k group: 16 bytes, a
// pixel's k-steps are consecutive 64-byte pieces of its row), all K of the group is held in registers, and the output
// channels are produced 64 at a time (weights as A, pixels as B: a lane finishes four consecutive channels of a pixel, the
// shared epilogue stores 16 bytes).  grid.y splits C_out when its weights exceed LDS (the input is then read once per split).
struct PwParams {
    const _Float16 *x, *scale, *w;
    ConvParams e;          // bias, extra, extra2, y, y2, K (= C_out), ldy, mode, slope: the shared epilogue's view
    long M;                // pixels
    int Kin, hw, n_per_wg; // input channels, pixels per image (scale index), output channels per workgroup
    const _Float16 *x2;    // NULL, or a SECOND input tensor whose channels continue the K axis after x's: y = act(W [x ; x2] + b ...) --
    int kt1;               // a residual block's last 1x1 and its 1x1 skip convolution as ONE product (k-steps of 32 taken from x: kt1)
    _Float16 *pool;        // NULL, or the 2x2 max-pool of the produced tensor (m / 4 pixels x C_out): groups are then 2-row blocks
    int W;                 // pixels per image row (pool mode)
};

// PT: 16-pixel tiles per wave and group (2: 32 pixels; 4: 64 pixels for the narrow inputs, whose groups are otherwise too small
// to pay for a group's fixed cost)
// EX: what the epilogue adds -- 0: nothing (mode 0), 1: `extra` (modes 1, 2), 2: the second output (modes 4, 5); POOL: the pooled
// output.  Compile-time, because a kernel that carries every form holds the registers of the richest one (the preloaded `extra` /
// `extra2` vectors, the pooled rows): the plain stream then spilled at 64 and at 448 / 512 input channels.  (Forcing 128 registers -- two workgroups per
// CU -- on the forms that need 130-136 was measured SLOWER, 0.39 against 0.33 ms for 256 -> 128 at 128 x 128 x 128.)
template <int KT, int PT, int EX, bool POOL>
__global__ __launch_bounds__(512) void k_pw(const PwParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_base = blockIdx.y * p.n_per_wg;
    const int Nw = p.e.K - n_base < p.n_per_wg ? p.e.K - n_base : p.n_per_wg;   // a multiple of 64
    constexpr int KG = KT * 4;   // 16-byte groups per weight row
    for (int i = threadIdx.x; i < Nw * KG; i += 512) {
        const int n = i / KG, q = i - n * KG;
        const int t = q >> 2, g = q & 3, r16 = n & 15, nt = n >> 4;
        const half8_t v = *reinterpret_cast<const half8_t *>(p.w + (long)(n_base + n) * (KT * 32) + q * 8);
        *reinterpret_cast<half8_t *>(smem + (nt * KT + t) * SUB + r16 * 64 + ((g * 16) ^ ((r16 >> 3) << 5))) = v;
    }
    __syncthreads();
    const int wfrag = (lane & 15) * 64 + (((lane >> 4) * 16) ^ (((lane & 15) >> 3) << 5));
    const int g = lane >> 4, pl = lane & 15;
    constexpr int GP = PT * 16;   // pixels per group
    constexpr int CB = PT >= 2 ? PT / 2 : 1;   // pool mode (PT >= 2): 16-pixel column blocks per group (the group is 2 rows x 16 CB columns)
    constexpr bool pooling = POOL;
    const long groups = (p.M + GP - 1) / GP;
    const int cblocks = pooling ? p.W / (16 * CB) : 1;   // groups per row pair
    for (long grp = (long)blockIdx.x * 8 + wave; grp < groups; grp += (long)gridDim.x * 8) {
        // pixel index of tile i's lane-0 pixel: consecutive pixels, or (pool mode) row 2 rp + i / CB, column block i % CB
        const long rp = pooling ? grp / cblocks : 0;
        const int cbk = pooling ? (int)(grp - rp * cblocks) : 0;
        auto tile_m = [&](int i) -> long {
            return pooling ? (2 * rp + i / CB) * p.W + (long)(cbk * CB + i % CB) * 16 : grp * GP + 16 * i;
        };
        const long m0 = tile_m(0);
        half8_t xf[PT][KT];
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const long m = tile_m(i) + pl;
            const long mm = m < p.M ? m : 0;
            const _Float16 *row = p.x + mm * (p.kt1 * 32) + g * 8;
            const _Float16 *row2 = p.x2 ? p.x2 + mm * ((KT - p.kt1) * 32) + g * 8 - p.kt1 * 32 : row;   // k-step t >= kt1: row2 + 32 t
#pragma unroll
            for (int t = 0; t < KT; t++) xf[i][t] = *reinterpret_cast<const half8_t *>((t < p.kt1 ? row : row2) + t * 32);
        }
        if (p.scale) {   // the SE gains of this group's image (a group never straddles two images: hw % 64 == 0 is asked of the caller)
            const _Float16 *sc = p.scale + (m0 / p.hw) * (KT * 32) + g * 8;
#pragma unroll
            for (int t = 0; t < KT; t++) {
                const half8_t sv = *reinterpret_cast<const half8_t *>(sc + t * 32);
#pragma unroll
                for (int i = 0; i < PT; i++) xf[i][t] = xf[i][t] * sv;
            }
        }
        for (int c0 = 0; c0 < Nw; c0 += 64) {
            float4_t acc[PT][4];
#pragma unroll
            for (int i = 0; i < PT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
            // the tensors the epilogue adds: requested BEFORE the matrix work of this 64-channel slice (16 bytes per lane and
            // (pixel tile, channel-tile pair): the lane's eight channels after the epilogue's lane exchange)
            half8_t pre[PT][2], pre2[PT][2];
            {
                const int gg = lane >> 4, odd = gg & 1, cbase = (gg & ~1) * 4;
#pragma unroll
                for (int i = 0; i < PT; i++) {
                    const long m = tile_m(i) + pl;
                    const long mm = m < p.M ? m : 0;
#pragma unroll
                    for (int jp = 0; jp < 2; jp++) {
                        const long o = mm * p.e.K + n_base + c0 + (2 * jp + odd) * 16 + cbase;
                        pre[i][jp] = pre2[i][jp] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                        if (EX == 1 || (EX == 2 && p.e.mode == 4)) pre[i][jp] = *reinterpret_cast<const half8_t *>(p.e.extra + o);
                        if (EX == 2) pre2[i][jp] = *reinterpret_cast<const half8_t *>(p.e.extra2 + o);
                    }
                }
            }
            const unsigned char *wb = smem + ((c0 >> 4) * KT) * SUB + wfrag;
#pragma unroll
            for (int t = 0; t < KT; t++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const half8_t wf = *reinterpret_cast<const half8_t *>(wb + (j * KT + t) * SUB);
#pragma unroll
                    for (int i = 0; i < PT; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[i][t], acc[i][j], 0, 0, 0);
                }
            }
            auto pix = [&](int i) -> long {
                const long m = tile_m(i) + pl;
                return m < p.M ? m : -1;
            };
            const long pb = pooling ? rp * (p.W >> 1) + (long)cbk * CB * 8 : 0;   // the group's first pooled pixel
            if (EX == 0) epilogue_preloaded<0, PT, 4, POOL>(acc, p.e, lane, n_base + c0, pix, pre, pre2, p.pool, pb);
            else if (EX == 1 && p.e.mode == 1) epilogue_preloaded<1, PT, 4, POOL>(acc, p.e, lane, n_base + c0, pix, pre, pre2, p.pool, pb);
            else if (EX == 1) epilogue_preloaded<2, PT, 4, POOL>(acc, p.e, lane, n_base + c0, pix, pre, pre2, p.pool, pb);
            else if (p.e.mode == 4) epilogue_preloaded<4, PT, 4, POOL>(acc, p.e, lane, n_base + c0, pix, pre, pre2, p.pool, pb);
            else epilogue_preloaded<5, PT, 4, POOL>(acc, p.e, lane, n_base + c0, pix, pre, pre2, p.pool, pb);
        }
    }
}

// pixels per group: 64 for the narrow inputs, 32 up to 512 channels, 16 beyond (all K of a group lives in registers: 32-pixel groups
// of 640 / 704 channels would spill 70-100 registers)
template <int KT, int PT, int EX, bool POOL>
int launch_pw_form(const PwParams &p, int n_split, hipStream_t st) {
    const int lds = p.n_per_wg * KT * 64;   // n_per_wg rows x (KT * 32) halves
    static const int inst = g_inst_count.fetch_add(1);
    if (const int rc = ensure_attr(reinterpret_cast<const void *>(&k_pw<KT, PT, EX, POOL>), lds, inst, st)) return rc;
    DevState *ds = dev_state();
    if (!ds) return PP_ERR_HIP;
    if (ds->ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return PP_ERR_HIP;
        ds->ncu = n >= 8 ? (n / 8) * 8 : 8;
    }
    // persistent grid: as many workgroups as are RESIDENT at once (registers and LDS decide: most forms hold one 8-wave workgroup
    // per CU, the leanest two) -- a workgroup beyond that would start when another ends and load the weights once more
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_pw<KT, PT, EX, POOL>), 512, (size_t)lds) !=
            hipSuccess || per_cu < 1)
        per_cu = 1;
    if (per_cu > 3) per_cu = 3;
    const long groups = (p.M + PT * 16 - 1) / (PT * 16);
    long gx = (long)ds->ncu * per_cu / n_split;
    if (gx < 1) gx = 1;
    if (gx * 8 > groups) gx = (groups + 7) / 8;
    hipLaunchKernelGGL((k_pw<KT, PT, EX, POOL>), dim3((unsigned)gx, (unsigned)n_split), dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}
template <int KT, int PT = (KT <= 2 ? 4 : (KT <= 16 ? 2 : 1))>
int launch_pw_inst(const PwParams &p, int n_split, hipStream_t st) {
    const int ex = p.e.mode == 0 ? 0 : (p.e.mode <= 2 ? 1 : 2);
    if (p.pool) {
        if constexpr (PT == 1) return PP_ERR_UNSUPPORTED;   // a 16-pixel group has no 2-row form
        else return ex == 0 ? launch_pw_form<KT, PT, 0, true>(p, n_split, st)
                   : ex == 1 ? launch_pw_form<KT, PT, 1, true>(p, n_split, st) : launch_pw_form<KT, PT, 2, true>(p, n_split, st);
    }
    return ex == 0 ? launch_pw_form<KT, PT, 0, false>(p, n_split, st)
         : ex == 1 ? launch_pw_form<KT, PT, 1, false>(p, n_split, st) : launch_pw_form<KT, PT, 2, false>(p, n_split, st);
}

// ------------------------------------------------------------------------------------------------ the stem: 7x7, stride 2
// models/layers_transposed.py:78-87 (Backbone.conv1 + bn1 + LeakyReLU): Conv2d(3, 64, 7, stride 2, padding 3) on the NHWC
// image.  MIOpen ran it as an implicit GEMM plus a separate bias / activation pass plus a layout copy (4.0 ms of a 176 ms
// step); its arithmetic is tiny (0.6 % of the forward), its output (2 B x 256 x 256 x 64 fp16 = 2.1 GB at 128 images) is what
// costs: it is HBM-bound, so this kernel is one pass -- read the image once, write the activation once.
// GEMM view: an output pixel's K = 7 rows x 8 input pixels x 3 channels = 168 (-> 192).  The window is taken EIGHT pixels wide
// (2 ox - 4 .. 2 ox + 3; the added left column has zero weights): 24 halves per row = three 16-byte groups, each a CONTIGUOUS
// run of the NHWC input row that starts on a 4-byte boundary -- the MFMA's pixel fragment (16 pixels x 4 groups of 8 k) is
// read straight from a linear LDS copy of the input rows, no im2col buffer.  Workgroup = 4 output rows x 64 columns, wave w
// takes row w: 64 pixels x 64 channels (4 x 4 tiles of v_mfma_f32_16x16x32_f16, weights as A / pixels as B as everywhere in
// this file), 6 k-steps.  LDS: 24 KB weights (sub-tiles of 16 channels x 32 k, swizzled like k_conv_igemm's) + 13 input rows.
constexpr int STEM_TW = 64, STEM_TH = 4, STEM_ROWS = 2 * STEM_TH + 5, STEM_K = 192;
constexpr int STEM_ROWB = ((2 * STEM_TW + 7) * 6 + 15) / 16 * 16 + 16;   // bytes of one staged input row (135 pixels, padded)

struct StemParams {
    const _Float16 *x, *w, *bias;   // x: (N, H, W, 3); w: (64, 192) k = (row, 8-pixel window, channel), prepared by the host
    _Float16 *y;                    // (N, H/2, W/2, 64)
    int N, H, W, Ho, Wo, tiles_x, tiles_y, tiles;
    float slope;
};

__global__ __launch_bounds__(256) void k_stem7x7(const StemParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *s_w = smem;                         // [4 channel tiles][6 k-steps][1 KiB]
    unsigned char *s_x = smem + 4 * 6 * SUB;           // [13][STEM_ROWB]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- weights -> LDS once per workgroup: sub-tile (ct, t) holds rows 16 ct .. + 15, k 32 t .. + 31, 64 B per row, swizzled
    for (int i = threadIdx.x; i < 64 * (STEM_K / 8); i += 256) {   // one 16-byte group per step
        const int n = i / (STEM_K / 8), q = i - n * (STEM_K / 8);
        const int t = q >> 2, g = q & 3, r16 = n & 15, ct = n >> 4;
        const half8_t v = *reinterpret_cast<const half8_t *>(p.w + (long)n * STEM_K + q * 8);
        *reinterpret_cast<half8_t *>(s_w + (ct * 6 + t) * SUB + r16 * 64 + ((g * 16) ^ ((r16 >> 3) << 5))) = v;
    }
    const int wfrag = (lane & 15) * 64 + (((lane >> 4) * 16) ^ (((lane & 15) >> 3) << 5));
    const int g = lane >> 4, pl = lane & 15;
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int n = tile / (p.tiles_x * p.tiles_y), rem = tile - n * (p.tiles_x * p.tiles_y);
        const int oy0 = (rem / p.tiles_x) * STEM_TH, ox0 = (rem % p.tiles_x) * STEM_TW;
        __syncthreads();   // the previous tile's fragment reads are done (and, first time, the weights are in place)
        // ---- input rows 2 oy0 - 3 .. 2 oy0 + 9, pixels 2 ox0 - 4 .. 2 ox0 + 130, as 8-byte units; outside the image: zeros
        const long row_bytes = (long)p.W * 6;
        const long first = (long)(2 * ox0 - 4) * 6;   // byte offset inside an input row; a multiple of 8 (ox0 is even)
        constexpr int UNITS = (2 * STEM_TW + 7) * 6 / 8 + 1;   // 102 units cover the 135 pixels
        for (int i = threadIdx.x; i < STEM_ROWS * UNITS; i += 256) {
            const int r = i / UNITS, u = i - r * UNITS;
            const int iy = 2 * oy0 - 3 + r;
            const long b = first + 8L * u;
            uint2 v = make_uint2(0u, 0u);
            if ((unsigned)iy < (unsigned)p.H && b >= 0 && b + 8 <= row_bytes)
                v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(p.x) + ((long)n * p.H + iy) * row_bytes + b);
            *reinterpret_cast<uint2 *>(s_x + r * STEM_ROWB + 8 * u) = v;
        }
        __syncthreads();
        float4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const int q = 4 * t + g;                       // this lane's 8-k group: input row q / 3, 16-byte group q % 3
            const int r = q < 21 ? q / 3 : 6, jg = q < 21 ? q % 3 : 0;   // groups 21..23 are padding (zero weights): read anything
            half8_t wf[4], xf[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wf[j] = *reinterpret_cast<const half8_t *>(s_w + (j * 6 + t) * SUB + wfrag);
            const unsigned char *rowp = s_x + (2 * wave + r) * STEM_ROWB + 16 * jg;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned *src = reinterpret_cast<const unsigned *>(rowp + 12 * (16 * i + pl));   // 4-byte aligned
                union {
                    unsigned u[4];
                    half8_t h;
                } cv;
                cv.u[0] = src[0], cv.u[1] = src[1], cv.u[2] = src[2], cv.u[3] = src[3];
                xf[i] = cv.h;
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        // ---- epilogue from registers: bias + LeakyReLU, 16-byte stores (epilogue_store)
        ConvParams e;
        e.bias = p.bias, e.extra = nullptr, e.extra2 = nullptr, e.y = p.y, e.y2 = nullptr;
        e.K = 64, e.ldy = 64, e.csum = nullptr, e.mode = 0, e.slope = p.slope;
        const int oy = oy0 + wave;
        epilogue_store<4, 4>(acc, e, lane, 0, [&](int i) -> long {
            const int ox = ox0 + 16 * i + pl;
            return (oy < p.Ho && ox < p.Wo) ? ((long)n * p.Ho + oy) * p.Wo + ox : -1;
        }, true);
    }
}

int launch_stem(const StemParams &p, hipStream_t st) {
    const int lds = 4 * 6 * SUB + STEM_ROWS * STEM_ROWB;
    static const int inst = g_inst_count.fetch_add(1);
    if (const int rc = ensure_attr(reinterpret_cast<const void *>(&k_stem7x7), lds, inst, st)) return rc;
    DevState *ds = dev_state();
    if (!ds) return PP_ERR_HIP;
    if (ds->ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return PP_ERR_HIP;
        ds->ncu = n >= 8 ? (n / 8) * 8 : 8;
    }
    const int grid = p.tiles < ds->ncu * 4 ? p.tiles : ds->ncu * 4;   // four workgroups per CU (36 KB of LDS each), each walks its tiles
    hipLaunchKernelGGL(k_stem7x7, dim3(grid), dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}

}  // namespace

extern "C" {

// y = act(conv1x1(x * scale[n]) + bias (+ extra)) [, y2 = y + extra2]: see k_pw.  x: DEVICE (m, c_in) fp16 (an NHWC activation,
// m = n * h * w pixels); scale: DEVICE (m / hw, c_in) fp16 or NULL; w: DEVICE (c_out, c_in); bias fp16[c_out]; extra / extra2 /
// y2: DEVICE (m, c_out) or NULL; y: DEVICE, pixel stride ldy >= c_out.  extra_mode as pp_conv_own_ex_f16 (0, 1, 2, 4).
// c_in in {64, 128, 192, 256, 384, 448, 512, 640, 704}, c_out % 64 == 0, hw % 64 == 0 when scale is given; PP_ERR_UNSUPPORTED otherwise.
PP_API int pp_pw_supported(int c_in, int c_out) {
    return ((c_in == 64 || c_in == 128 || c_in == 192 || c_in == 256 || c_in == 384 || c_in == 448 || c_in == 512 || c_in == 640 || c_in == 704) && c_out % 64 == 0 && c_out >= 64) ? 1 : 0;
}
static int pw_run(const void *x, const void *scale, const void *w, const void *bias, const void *extra, const void *extra2, void *y,
                  void *y2, long m, int hw, int c_in, int c_out, int ldy, int extra_mode, float slope, void *pool, int width, void *stream,
                  const void *x2 = nullptr, int c_in2 = 0);

PP_API int pp_pw_f16(const void *x, const void *scale, const void *w, const void *bias, const void *extra, const void *extra2, void *y,
                     void *y2, long m, int hw, int c_in, int c_out, int ldy, int extra_mode, float slope, void *stream) {
    return pw_run(x, scale, w, bias, extra, extra2, y, y2, m, hw, c_in, c_out, ldy, extra_mode, slope, nullptr, 0, stream);
}

// The same with the 2x2 / stride 2 max-pool of the produced tensor (y; y2 in mode 4) as one more output: pool_out DEVICE
// (m / 4, c_out) = (n, h / 2, w / 2, c_out); width = w (pixels per row), a multiple of 32 (64 for c_in = 64), h even.
PP_API int pp_pw_pool_f16(const void *x, const void *scale, const void *w, const void *bias, const void *extra, const void *extra2,
                          void *y, void *y2, void *pool_out, long m, int hw, int width, int c_in, int c_out, int ldy, int extra_mode,
                          float slope, void *stream) {
    if (!pool_out || width <= 0 || hw % width || ((hw / width) & 1) || (reinterpret_cast<uintptr_t>(pool_out) & 15)) return PP_ERR_BAD_ARG;
    if (width % (c_in == 64 ? 64 : 32) || m % hw) return PP_ERR_UNSUPPORTED;
    return pw_run(x, scale, w, bias, extra, extra2, y, y2, m, hw, c_in, c_out, ldy, extra_mode, slope, pool_out, width, stream);
}

// [x ; x2]: y = act(W [x ; x2] + bias ...), W: (c_out, c_in + c_in2) -- e.g. a residual block's last 1x1 convolution and its 1x1 skip
// convolution (models/layers_transposed.py:12-48: out = conv3(t) + skip(x)) as one product, the skip's output never materialised.
// pool_out / width as pp_pw_pool_f16 (NULL / 0: none).  c_in, c_in2 multiples of 32 with a supported sum.
PP_API int pp_pw_cat_f16(const void *x, const void *x2, const void *w, const void *bias, const void *extra, void *y, void *pool_out, long m,
                         int hw, int width, int c_in, int c_in2, int c_out, int ldy, int extra_mode, float slope, void *stream) {
    if (!x2 || c_in <= 0 || c_in2 <= 0 || (c_in & 31) || (c_in2 & 31) || (reinterpret_cast<uintptr_t>(x2) & 15) || extra_mode > 2) return PP_ERR_BAD_ARG;
    if (pool_out && (width <= 0 || hw % width || ((hw / width) & 1) || width % (c_in + c_in2 == 64 ? 64 : 32) || m % hw)) return PP_ERR_UNSUPPORTED;
    return pw_run(x, nullptr, w, bias, extra, nullptr, y, nullptr, m, hw, c_in + c_in2, c_out, ldy, extra_mode, slope, pool_out,
                  pool_out ? width : 0, stream, x2, c_in2);
}

static int pw_run(const void *x, const void *scale, const void *w, const void *bias, const void *extra, const void *extra2, void *y,
                  void *y2, long m, int hw, int c_in, int c_out, int ldy, int extra_mode, float slope, void *pool, int width, void *stream,
                  const void *x2, int c_in2) {
    // extra_mode 5 (this kernel only): the second output y2 = y + extra2 WITHOUT a tensor added before the activation
    if (!x || !w || !bias || !y || m <= 0 || hw <= 0 || ldy < c_out || extra_mode < 0 || extra_mode == 3 || extra_mode > 5 ||
        (extra_mode != 0 && extra_mode != 5) != (extra != nullptr) || (extra_mode >= 4) != (extra2 != nullptr) ||
        (extra_mode >= 4) != (y2 != nullptr))
        return PP_ERR_BAD_ARG;
    if (!pp_pw_supported(c_in, c_out) || !(slope >= 0.f && slope <= 1.f) || (ldy & 7)) return PP_ERR_UNSUPPORTED;
    if (scale && (hw & 63)) return PP_ERR_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(bias) |
                         reinterpret_cast<uintptr_t>(extra) | reinterpret_cast<uintptr_t>(extra2) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(y2) | reinterpret_cast<uintptr_t>(scale);
    if (al & 15) return PP_ERR_BAD_ARG;
    PwParams p;
    p.x = static_cast<const _Float16 *>(x);
    p.scale = static_cast<const _Float16 *>(scale);
    p.w = static_cast<const _Float16 *>(w);
    p.e.bias = static_cast<const _Float16 *>(bias);
    p.e.extra = static_cast<const _Float16 *>(extra);
    p.e.extra2 = static_cast<const _Float16 *>(extra2);
    p.e.y = static_cast<_Float16 *>(y);
    p.e.y2 = static_cast<_Float16 *>(y2);
    p.e.K = c_out, p.e.ldy = ldy, p.e.mode = extra_mode, p.e.slope = slope;
    p.M = m, p.Kin = c_in, p.hw = hw;
    p.pool = static_cast<_Float16 *>(pool), p.W = width;
    p.x2 = static_cast<const _Float16 *>(x2), p.kt1 = (c_in - c_in2) / 32;   // c_in is the TOTAL here
    // output channels per workgroup: all of them when their weights fit LDS (144 KiB), else the fewest equal splits that do
    int n_split = 1;
    while ((c_out / n_split) * c_in * 2 > 144 * 1024 || c_out % n_split || (c_out / n_split) % 64) {
        if (++n_split > c_out / 64) return PP_ERR_UNSUPPORTED;
    }
    p.n_per_wg = c_out / n_split;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (c_in / 32) {
        case 2: return launch_pw_inst<2>(p, n_split, st);
        case 4: return launch_pw_inst<4>(p, n_split, st);
        case 6: return launch_pw_inst<6>(p, n_split, st);
        case 8: return launch_pw_inst<8>(p, n_split, st);
        case 12: return launch_pw_inst<12>(p, n_split, st);
        case 14: return launch_pw_inst<14>(p, n_split, st);   // 192 + 256: the [t ; x] input of the first hourglass level's residual blocks
        case 16: return launch_pw_inst<16>(p, n_split, st);
        case 20: return launch_pw_inst<20>(p, n_split, st);   // 256 + 384, 320 + ... : the second hourglass level's [t ; x]
        case 22: return launch_pw_inst<22>(p, n_split, st);
        default: return PP_ERR_UNSUPPORTED;
    }
}

// The stem of the IMHN (models/layers_transposed.py:78-87): y = leaky(conv7x7_stride2_pad3(x, w) + bias) on the NHWC image.
// x: DEVICE (n, h, w, 3) fp16 with h, w even and w % 4 == 0; w_prepared: DEVICE (64, 192) fp16 in the kernel's k order -- row r
// (7), window pixel s' (8; s' = 0 is a zero column, s' = s + 1), channel c (3), then zeros up to 192 -- i.e.
// w_prepared[k][(r * 8 + s') * 3 + c] = weight[k][c][r][s' - 1]; bias fp16[64]; y: DEVICE (n, h / 2, w / 2, 64).
PP_API int pp_stem7x7_f16(const void *x, const void *w_prepared, const void *bias, void *y, int n, int h, int wd, float slope, void *stream) {
    if (!x || !w_prepared || !bias || !y || n <= 0 || h <= 0 || wd <= 0) return PP_ERR_BAD_ARG;
    if ((h & 1) || (wd & 3) || !(slope >= 0.f && slope <= 1.f)) return PP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_prepared) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y)) & 15)
        return PP_ERR_BAD_ARG;
    StemParams p;
    p.x = static_cast<const _Float16 *>(x);
    p.w = static_cast<const _Float16 *>(w_prepared);
    p.bias = static_cast<const _Float16 *>(bias);
    p.y = static_cast<_Float16 *>(y);
    p.N = n, p.H = h, p.W = wd, p.Ho = h / 2, p.Wo = wd / 2;
    p.tiles_x = (p.Wo + STEM_TW - 1) / STEM_TW;
    p.tiles_y = (p.Ho + STEM_TH - 1) / STEM_TH;
    const long tiles = (long)n * p.tiles_x * p.tiles_y;
    if (tiles >= (1L << 31)) return PP_ERR_TOO_LARGE;
    p.tiles = (int)tiles;
    p.slope = slope;
    return launch_stem(p, static_cast<hipStream_t>(stream));
}

// 1 when pp_conv_own_f16 takes the shape: stride 1, square kernel, C_in % 32 == 0, C_out % 64 == 0
// Diagnostics (POSEPAF_CONV_DBG with bit 1024): median over workgroups of the shader-clock cycles a workgroup spent per section,
// summed over its tiles, in the last stamped launch.  out[0..4] = wait for first DMA + stores / first fragment reads / main loop /
// next tile's decode + DMA issue / epilogue; out[5] = main loop in 100 MHz ticks (clock in GHz = out[2] / out[5] / 10).
PP_API int pp_conv_debug_clock(double *out, int nwg) {
    static unsigned long long h[8 * 512];
    if (!out || nwg <= 0 || nwg > 512) return PP_ERR_BAD_ARG;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_conv_clk), sizeof(h)) != hipSuccess) return PP_ERR_HIP;
    for (int k = 0; k < 6; k++) {   // median over workgroups of each section's cycle sum
        double v[512];
        for (int i = 0; i < nwg; i++) v[i] = (double)h[i * 8 + k];
        for (int i = 1; i < nwg; i++) {
            const double x = v[i];
            int j = i - 1;
            for (; j >= 0 && v[j] > x; j--) v[j + 1] = v[j];
            v[j + 1] = x;
        }
        out[k] = v[nwg / 2];
    }
    return PP_OK;
}

PP_API int pp_conv_own_supported(int c_in, int c_out, int ksize) { return (c_in % 32 == 0 && c_out % 64 == 0 && ksize >= 1 && ksize <= 7) ? 1 : 0; }

static int conv_own_run(const void *x, const void *w, const void *bias, const void *extra, const void *extra2, void *y, void *y2,
                        int n, int h, int wd, int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int bn,
                        int upsampled_input, int ldx, int ldy, void *stream);

PP_API int pp_conv_own_ex_f16(const void *x, const void *w, const void *bias, const void *extra, const void *extra2, void *y, void *y2,
                       int n, int h, int wd, int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int bn,
                       int upsampled_input, void *stream) {
    return conv_own_run(x, w, bias, extra, extra2, y, y2, n, h, wd, c_in, c_out, ksize, pad, dilation, extra_mode, slope, bn,
                        upsampled_input, c_in, c_out, stream);
}

// pp_conv_own_f16 on channel SLICES of wider NHWC tensors: ldx / ldy = elements between consecutive pixels of x / y (>= c_in /
// c_out, multiples of 8).  The 3x3 halo kernel only (bn = 512; pad = dilation = 1, or 3 / 4 / 5): PP_ERR_UNSUPPORTED for the shapes
// it does not take.  `extra` stays packed (n, h, w, c_out).
PP_API int pp_conv_own_ld_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd, int c_in,
                              int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int bn, int ldx, int ldy,
                              void *stream) {
    if (extra_mode > 2 || ldx < c_in || ldy < c_out || ((ldx | ldy) & 7)) return PP_ERR_BAD_ARG;
    if (bn != 512) return PP_ERR_UNSUPPORTED;
    return conv_own_run(x, w, bias, extra, nullptr, y, nullptr, n, h, wd, c_in, c_out, ksize, pad, dilation, extra_mode, slope, bn, 0,
                        ldx, ldy, stream);
}

static int conv_own_run(const void *x, const void *w, const void *bias, const void *extra, const void *extra2, void *y, void *y2,
                        int n, int h, int wd, int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int bn,
                        int upsampled_input, int ldx, int ldy, void *stream) {
    if (!x || !w || !bias || !y || n <= 0 || h <= 0 || wd <= 0 || ksize <= 0 || pad < 0 || dilation <= 0 || extra_mode < 0 ||
        extra_mode > 4 || (extra_mode != 0) != (extra != nullptr) || (extra_mode >= 3) != (extra2 != nullptr) ||
        (extra_mode == 4) != (y2 != nullptr) || (upsampled_input != 0 && upsampled_input != 1) ||
        ((reinterpret_cast<uintptr_t>(extra2) | reinterpret_cast<uintptr_t>(y2)) & 15))
        return PP_ERR_BAD_ARG;
    if ((upsampled_input || extra_mode == 3) && bn != 512) return PP_ERR_UNSUPPORTED;   // the 8-wave 3x3 halo kernel only
    if (upsampled_input && ((h | wd) & 1)) return PP_ERR_BAD_ARG;
    if (!pp_conv_own_supported(c_in, c_out, ksize)) return PP_ERR_UNSUPPORTED;
    if (!(slope >= 0.f && slope <= 1.f)) return PP_ERR_UNSUPPORTED;   // the epilogue's LeakyReLU is max(t, slope * t)
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(bias) |
                         reinterpret_cast<uintptr_t>(extra) | reinterpret_cast<uintptr_t>(y);
    if (al & 15) return PP_ERR_BAD_ARG;
    const int ho = h + 2 * pad - dilation * (ksize - 1), wo = wd + 2 * pad - dilation * (ksize - 1);
    if (ho <= 0 || wo <= 0) return PP_ERR_BAD_ARG;
    if (bn != 0 && bn != 512 && ((bn != 256 && bn != 128 && bn != 64) || c_out % bn)) return PP_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DevState *ds = dev_state();
    if (!ds) return PP_ERR_HIP;
    if (!ds->zero) {
        if (capturing(st)) return PP_ERR_UNSUPPORTED;   // first use on this device allocates: not inside a capture
        if (hipMalloc(&ds->zero, 256) != hipSuccess || hipMemset(ds->zero, 0, 256) != hipSuccess) return PP_ERR_HIP;
    }
    ConvParams p;
    p.x = static_cast<const _Float16 *>(x);
    p.w = static_cast<const _Float16 *>(w);
    p.bias = static_cast<const _Float16 *>(bias);
    p.extra = static_cast<const _Float16 *>(extra);
    p.extra2 = static_cast<const _Float16 *>(extra2);
    p.up = upsampled_input;
    p.y = static_cast<_Float16 *>(y);
    p.y2 = static_cast<_Float16 *>(y2);
    p.zero = ds->zero;
    p.N = n; p.H = h; p.W = wd; p.C = c_in; p.K = c_out; p.R = ksize; p.pad = pad; p.dil = dilation; p.Ho = ho; p.Wo = wo;
    p.ldy = ldy;
    p.ldx = ldx;
    p.csum = nullptr;
    p.pad_y = p.pad_x = pad, p.up_out = 0, p.py = p.px = 0;
    p.M = (long)n * ho * wo;
    p.mode = extra_mode;
    p.slope = slope;
#ifdef PP_CONV_DIAG   // the diagnostics library only (make diag): compile-time ablated instances behind POSEPAF_CONV_DBG
    static const int dbg = std::getenv("POSEPAF_CONV_DBG") ? std::atoi(std::getenv("POSEPAF_CONV_DBG")) : 0;
    p.dbg = dbg;
#else
    p.dbg = 0;            // the product library has no ablated instance and reads no such variable
#endif
    static const int stagger = std::getenv("POSEPAF_CONV_STAGGER") ? std::atoi(std::getenv("POSEPAF_CONV_STAGGER")) : -1;   // -1: launcher's default
    p.stagger = stagger;
    if (bn == 0 || bn == 512) {   // 512: the halo-tile 3x3 kernel
        HaloParams g;
        if (halo_geometry(p, g)) return launch_halo(p, g, st);
        if (bn != 0 || ldx != c_in || ldy != c_out) return PP_ERR_UNSUPPORTED;   // (the implicit-GEMM kernel reads packed pixels)
        bn = c_out % 256 == 0 ? 256 : (c_out % 128 == 0 ? 128 : 64);
    }
    switch (bn) {
        case 256: return launch<256>(p, st);
        case 128: return launch<128>(p, st);
        default: return launch<64>(p, st);
    }
}

// The 3x3 / pad 1 halo kernel with the SE squeeze of its output on the side: y = act(conv + bias) and sums_ws[image][split][c_out] =
// partial sums of the binary16 outputs over the pixels of split (tile, pixel wave), fp32; pp_conv_own_sums_splits(h, wd) gives the
// number of splits per image (0: the shape is not taken).  pp_channel_mean_finish_f16 turns the partials into the mean.
PP_API int pp_conv_own_sums_splits(int h, int wd) {
    ConvParams q;
    q.R = 3, q.pad = 1, q.dil = 1, q.C = 32, q.K = 128, q.H = h, q.W = wd, q.N = 1, q.csum = nullptr;
    HaloParams g;
    return (halo_geometry(q, g) && g.gimg == 1) ? g.tiles * 4 : 0;
}
PP_API int pp_conv_own_sums_f16(const void *x, const void *w, const void *bias, void *y, void *sums_ws, int n, int h, int wd, int c_in,
                                int c_out, float slope, void *stream) {
    if (!x || !w || !bias || !y || !sums_ws || n <= 0 || h <= 0 || wd <= 0) return PP_ERR_BAD_ARG;
    if (!pp_conv_own_supported(c_in, c_out, 3) || !(slope >= 0.f && slope <= 1.f)) return PP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y) |
         reinterpret_cast<uintptr_t>(sums_ws)) & 15)
        return PP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvParams p;
    p.x = static_cast<const _Float16 *>(x), p.w = static_cast<const _Float16 *>(w), p.bias = static_cast<const _Float16 *>(bias);
    p.extra = nullptr, p.extra2 = nullptr, p.up = 0, p.y = static_cast<_Float16 *>(y), p.y2 = nullptr, p.zero = nullptr;
    p.N = n, p.H = h, p.W = wd, p.C = c_in, p.K = c_out, p.R = 3, p.pad = 1, p.dil = 1, p.Ho = h, p.Wo = wd;
    p.ldy = c_out, p.csum = static_cast<float *>(sums_ws), p.M = (long)n * h * wd, p.mode = 0, p.slope = slope, p.dbg = 0, p.stagger = -1;
    p.pad_y = p.pad_x = 1, p.up_out = 0, p.py = p.px = 0;
    HaloParams g;
    if (!halo_geometry(p, g)) return PP_ERR_UNSUPPORTED;
    return launch_halo(p, g, st);
}

// conv3x3 / pad 1 behind a x2 nearest-neighbour upsample (the hourglass' `hg[i][3](upsample(low))`, models/layers_transposed.py:
// 270-275) WITHOUT the 2.25x redundant arithmetic: an output pixel (2 y + py, 2 x + px) sees only a 2 x 2 neighbourhood of the
// half-resolution input, each input pixel through the SUM of the 3x3 taps that land on it --
//   rows:  py = 0: x[y - 1] * w[0] + x[y] * (w[1] + w[2])      py = 1: x[y] * (w[0] + w[1]) + x[y + 1] * w[2]     (columns alike)
// -- so the layer is four 2x2 convolutions of the half-resolution tensor (one per output phase, 16 instead of 36 multiply-adds per
// input pixel), run here as four launches of the implicit-GEMM kernel with asymmetric tap offsets and an interleaving output
// map.  w4: DEVICE [4 phases: (py, px) = (0,0), (0,1), (1,0), (1,1)][c_out][2][2][c_in] fp16, the tap sums formed by the caller in
// fp32 and rounded once (same real-number result as the 3x3 on the upsampled tensor; the fp16 rounding of the weights differs).
// x: DEVICE (n, h_low, w_low, c_in); y / extra / extra2: DEVICE (n, 2 h_low, 2 w_low, c_out).  extra_mode 0, 2 or 3 as
// pp_conv_own_ex_f16; bn = 256 / 128 / 64 (0: the largest that divides c_out).
PP_API int pp_conv_up2_collapsed_f16(const void *x, const void *w4, const void *bias, const void *extra, const void *extra2, void *y, int n,
                                     int h_low, int w_low, int c_in, int c_out, int extra_mode, float slope, int bn, void *stream) {
    if (!x || !w4 || !bias || !y || n <= 0 || h_low <= 0 || w_low <= 0 || (extra_mode != 0 && extra_mode != 2 && extra_mode != 3) ||
        (extra_mode != 0) != (extra != nullptr) || (extra_mode == 3) != (extra2 != nullptr))
        return PP_ERR_BAD_ARG;
    if (!pp_conv_own_supported(c_in, c_out, 2) || !(slope >= 0.f && slope <= 1.f)) return PP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w4) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(extra) |
         reinterpret_cast<uintptr_t>(extra2) | reinterpret_cast<uintptr_t>(y)) & 15)
        return PP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bn == 512) {   // the halo-tile kernel walking only the four taps each output phase sees (k_conv3x3_halo<..., PH>)
        ConvParams q;
        q.x = static_cast<const _Float16 *>(x), q.bias = static_cast<const _Float16 *>(bias);
        q.extra = static_cast<const _Float16 *>(extra), q.extra2 = static_cast<const _Float16 *>(extra2);
        q.y = static_cast<_Float16 *>(y), q.y2 = nullptr, q.zero = nullptr, q.up = 0;
        q.N = n, q.H = h_low, q.W = w_low, q.C = c_in, q.K = c_out, q.R = 3, q.pad = 1, q.dil = 1, q.Ho = h_low, q.Wo = w_low;
        q.ldy = c_out, q.csum = nullptr, q.M = (long)n * h_low * w_low, q.mode = extra_mode, q.slope = slope, q.dbg = 0, q.stagger = -1;
        q.pad_y = q.pad_x = 1, q.up_out = 1, q.py = q.px = 0;
        HaloParams g;
        if (!halo_geometry(q, g)) return PP_ERR_UNSUPPORTED;
        const long wphase = (long)c_out * 4 * c_in;
        for (int ph = 0; ph < 4; ph++) {
            q.w = static_cast<const _Float16 *>(w4) + ph * wphase;
            const int rc = ph == 0 ? launch_halo_phase<0>(q, g, st) : (ph == 1 ? launch_halo_phase<1>(q, g, st) :
                           (ph == 2 ? launch_halo_phase<2>(q, g, st) : launch_halo_phase<3>(q, g, st)));
            if (rc != PP_OK) return rc;
        }
        return PP_OK;
    }
    if (bn == 0) bn = c_out % 256 == 0 ? 256 : (c_out % 128 == 0 ? 128 : 64);
    if ((bn != 256 && bn != 128 && bn != 64) || c_out % bn) return PP_ERR_UNSUPPORTED;
    DevState *ds = dev_state();
    if (!ds) return PP_ERR_HIP;
    if (!ds->zero) {
        if (capturing(st)) return PP_ERR_UNSUPPORTED;
        if (hipMalloc(&ds->zero, 256) != hipSuccess || hipMemset(ds->zero, 0, 256) != hipSuccess) return PP_ERR_HIP;
    }
    ConvParams p;
    p.x = static_cast<const _Float16 *>(x), p.bias = static_cast<const _Float16 *>(bias);
    p.extra = static_cast<const _Float16 *>(extra), p.extra2 = static_cast<const _Float16 *>(extra2);
    p.y = static_cast<_Float16 *>(y), p.y2 = nullptr, p.zero = ds->zero, p.up = 0;
    p.N = n, p.H = h_low, p.W = w_low, p.C = c_in, p.K = c_out, p.R = 2, p.pad = 0, p.dil = 1, p.Ho = h_low, p.Wo = w_low;
    p.ldy = c_out, p.csum = nullptr, p.M = (long)n * h_low * w_low, p.mode = extra_mode, p.slope = slope, p.dbg = 0, p.stagger = -1;
    p.up_out = 1;
    const long wphase = (long)c_out * 4 * c_in;   // halves per phase
    for (int ph = 0; ph < 4; ph++) {
        p.py = ph >> 1, p.px = ph & 1;
        p.pad_y = 1 - p.py, p.pad_x = 1 - p.px;   // taps reach rows {y - 1, y} (py = 0) or {y, y + 1} (py = 1); columns alike
        p.w = static_cast<const _Float16 *>(w4) + ph * wphase;
        const int rc = bn == 256 ? launch<256>(p, st) : (bn == 128 ? launch<128>(p, st) : launch<64>(p, st));
        if (rc != PP_OK) return rc;
    }
    return PP_OK;
}

PP_API int pp_conv_own_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd, int c_in,
                    int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int bn, void *stream) {
    if (extra_mode > 2) return PP_ERR_BAD_ARG;
    return pp_conv_own_ex_f16(x, w, bias, extra, nullptr, y, nullptr, n, h, wd, c_in, c_out, ksize, pad, dilation, extra_mode, slope, bn, 0,
                              stream);
}

}  // extern "C"
