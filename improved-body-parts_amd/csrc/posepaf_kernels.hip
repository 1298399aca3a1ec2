// posepaf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the bottom-up pose post-processing
// path.  TWO launches per batch of images, every input byte read from HBM exactly once:
//
//   K_A  k_heat_peaks    grid (18, B)  one 256-thread workgroup per keypoint channel
//        flip-average the channel straight into LDS (A2), plus-shaped NMS with row-major ordered compaction
//        by wave ballots (A3), x4 bicubic patch refinement + wave arg-max per peak (A4); its last workgroup
//        ranks the images by matching load (dispatch order of K_B)
//   K_B  k_limb_connect  grid (31, B)  workgroups 0..29 of an image: one limb channel each
//        flip-average the limb map into LDS rows padded by 16 bytes (A2), line-integral scoring of every (a,b)
//        peak pair with the x4 bicubic evaluated on the fly at the sampled pixel instead of materialising the
//        31.5 MB up-sampled map (A4'+A5; sparse limbs spread the SAMPLES of a pair over lanes), ordered
//        compaction of accepted candidates, the reference's sort order and greedy matching (A6), publication
//        of the limb through a per-limb flag;
//        workgroup 30 of an image: ONE wave that assembles the image (A7) WHILE its limbs are matched -- it
//        consumes limb 0, 1, ... as each is published, prunes, and writes the fixed-size result record
//   (k_assemble_wave / k_assemble: the assembly as its own launch -- diagnostics, the drop-in process_paf and the
//   Python-twin path; k_accumulate_scales + k_fullres_peaks + ...: the original path at image resolution.)
//
// No MFMA anywhere: this is gather/compare/reduce work bounded by HBM.
// Floating point: every expression mirrors the reference's operation ORDER and rounding (x86-64 g++ without
// FMA): products and sums are kept separate with __fmul_rn/__fadd_rn and the file is built with
// -ffp-contract=off; divisions and sqrt are the correctly rounded forms (hipcc default).
//
// file:line citations are relative to /root/reference.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "posepaf_internal.h"

namespace pp {

// ------------------------------------------------------------------------------------------------ tables
// utils/pafprocess/pafprocess.h:21-27 (== config/config.py:114-121)
__device__ const int8_t d_limb_pairs[PP_NUM_LIMB][2] = {
    {1, 0},   {1, 14},  {1, 15},  {1, 16},  {1, 17},  {0, 14},  {0, 15},  {14, 16}, {15, 17}, {1, 2},
    {2, 3},   {3, 4},   {1, 5},   {5, 6},   {6, 7},   {1, 8},   {8, 9},   {9, 10},  {1, 11},  {11, 12},
    {12, 13}, {0, 2},   {0, 5},   {2, 8},   {8, 12},  {5, 11},  {11, 9},  {16, 2},  {17, 5},  {8, 11}};
constexpr int kLimbA[PP_NUM_LIMB] = {1, 1, 1, 1, 1, 0, 0, 14, 15, 1, 2, 3, 1, 5, 6, 1, 8, 9, 1, 11, 12, 0, 0, 2, 8, 5, 11, 16, 17, 8};
constexpr int kLimbB[PP_NUM_LIMB] = {0, 14, 15, 16, 17, 14, 15, 16, 17, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 2, 5, 8, 12, 11, 9, 2, 5, 11};
// config/config.py:150-152
__device__ const int8_t d_flip_heat_ord[PP_NUM_HEAT] = {0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16, 18, 19};
__device__ const int8_t d_flip_paf_ord[PP_NUM_LIMB] = {0,  2,  1,  4,  3,  6,  5,  8,  7,  12, 13, 14, 9,  10, 11,
                                                       18, 19, 20, 15, 16, 17, 22, 21, 25, 26, 23, 24, 28, 27, 29};

// OpenCV interpolateCubic(A = -0.75) at the four fractional phases of an x4 upsample:
// fx = (dx + 0.5)/4 - 0.5  ->  frac = .625, .875, .125, .375 for dx mod 4 = 0..3.  All sixteen values are exact
// dyadic rationals in binary32 (no rounding in their derivation), so a literal table is bit-identical to the
// float computation the oracle performs.
__device__ const float d_cubic4[4][4] = {{-0.06591796875f, 0.42626953125f, 0.74951171875f, -0.10986328125f},
                                         {-0.01025390625f, 0.11474609375f, 0.96728515625f, -0.07177734375f},
                                         {-0.07177734375f, 0.96728515625f, 0.11474609375f, -0.01025390625f},
                                         {-0.10986328125f, 0.74951171875f, 0.42626953125f, -0.06591796875f}};

// Diagnostic phase stamps (shader clock) written per workgroup when a stamp buffer is registered with
// pp_debug_set_stamps(); a null pointer (the default) costs one scalar branch per phase.
__device__ long long *d_stamps = nullptr;
__device__ int d_stamp_realtime = 0;
__device__ int d_sp_max_pairs = 128;   // K_B: limbs with at most this many candidate pairs spread a pair's SAMPLES over lanes (<= 512)   // 1: the chip-wide 100 MHz counter (timelines across CUs) instead of the shader clock
__device__ __forceinline__ void stamp(long long *buf, int wg, int slot) {
    if (buf && threadIdx.x == 0) buf[(size_t)wg * 8 + slot] = d_stamp_realtime ? (long long)wall_clock64() : (long long)clock64();
}

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float ldsf(const float *p, int i) { return p[i]; }
__device__ __forceinline__ float ldsf(const __half *p, int i) { return __half2float(p[i]); }
// 8 consecutive map values starting at a multiple of 8 (16-byte aligned for binary16, two 16-byte reads for f32)
__device__ __forceinline__ void load8(const __half *p, float out[8]) {
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    const __half2 *h2 = reinterpret_cast<const __half2 *>(&q);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 f = __half22float2(h2[j]);
        out[2 * j] = f.x;
        out[2 * j + 1] = f.y;
    }
}
__device__ __forceinline__ void load8(const float *p, float out[8]) {
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
    out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
}
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// Write-through (sc1) stores for data that ANOTHER workgroup of the same launch reads (K_B's connections, read by the
// workgroup that assembles the image): the bytes leave the XCD's L2 with the store, so the publisher needs no agent-scope
// release fence -- every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane
// draws the ticket; the reader acquires (cdna_hip_programming.md Guideline 16, form R1).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(float4 *p, const float4 &v) {
    const f32x4_t q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void store_sc1(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_sc1(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Per-image status flags: one word per producing workgroup -- [0, 18) written by the peak kernel of each part, [18, 48) by
// the limb kernel of each limb -- stored with a PLAIN store by every workgroup on every launch (early exits included), and
// OR-ed by the assembly.  Nothing is ever zeroed by a memset and nothing is accumulated with atomics, so no word can
// carry anything but the flags of the launch that produced it.  `first` skips the part words on the host-array paths
// (process_paf / find_connections), where the peaks come from the caller and no peak kernel ran.
constexpr int kFlagWords = PP_NUM_PART + PP_NUM_LIMB;
__device__ __forceinline__ unsigned or_flags(const unsigned *flags, int img, int first, int lane) {
    unsigned v = (lane >= first && lane < kFlagWords) ? flags[img * kFlagWords + lane] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d);
    return v;
}

// ------------------------------------------------------------------------------------------------ A2 loader
// utils/parse_skeletons.py:82-103: avg[c][y][x] = (o0[c][y][x] + o1[perm(c)][y][w-1-x]) / 2 computed in the
// array's dtype (binary16 add + binary16 halve for the AMP output, float32 otherwise), then widened.
// The binary16 average is itself a binary16 value, so the LDS copy is kept in binary16 for that dtype
// (half the LDS footprint, exact).  Coalesced 16-byte reads; the mirrored operand is read as the mirrored
// 16-byte vector and reversed in registers.
__device__ __forceinline__ __half2 avg_h2(__half2 a, __half2 b) {
    return __hmul2(__hadd2(a, b), __float2half2_rn(0.5f));
}
__device__ __forceinline__ __half2 swap_h2(__half2 v) { return __lowhigh2highlow(v); }

// Each thread issues kLoadUnroll 16-byte loads per operand before touching any of them (2*kLoadUnroll loads in
// flight per lane) -- the loop is latency-bound otherwise.
constexpr int kLoadUnroll = 4;

// NT = threads of the calling workgroup; ld = LDS row stride in elements (ld == w: dense; K_B pads every row by 16 bytes so
// that equal columns of neighbouring rows fall into different banks -- its gathers run along near-vertical limbs).
template <int NT = kThreads>
__device__ void load_channel(__half *smap, const __half *o0, const __half *o1, int h, int w, bool flip, int ld = 0) {
    if (ld == 0) ld = w;
    const int npix = h * w;
    const bool vec_ok = (w % 8 == 0) && ((reinterpret_cast<uintptr_t>(o0) & 15) == 0) &&
                        (!flip || (reinterpret_cast<uintptr_t>(o1) & 15) == 0);
    if (vec_ok) {
        const int nvec = npix / 8;
        const int vpr = w / 8;  // vectors per row
        const uint4 *p0 = reinterpret_cast<const uint4 *>(o0);
        const uint4 *p1 = reinterpret_cast<const uint4 *>(o1);
        for (int v0 = threadIdx.x; v0 < nvec; v0 += NT * kLoadUnroll) {
            uint4 a[kLoadUnroll], m[kLoadUnroll];
#pragma unroll
            for (int u = 0; u < kLoadUnroll; u++) {
                const int v = v0 + u * NT;
                if (v < nvec) {
                    a[u] = p0[v];
                    if (flip) {
                        const int y = v / vpr, vx = v - y * vpr;
                        m[u] = p1[y * vpr + (vpr - 1 - vx)];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kLoadUnroll; u++) {
                const int v = v0 + u * NT;
                if (v < nvec) {
                    if (flip) {
                        __half2 *ah = reinterpret_cast<__half2 *>(&a[u]);
                        const __half2 *mh = reinterpret_cast<const __half2 *>(&m[u]);
                        // reversed 8-vector: element j pairs with mirrored element 7-j
                        ah[0] = avg_h2(ah[0], swap_h2(mh[3]));
                        ah[1] = avg_h2(ah[1], swap_h2(mh[2]));
                        ah[2] = avg_h2(ah[2], swap_h2(mh[1]));
                        ah[3] = avg_h2(ah[3], swap_h2(mh[0]));
                    }
                    const int yy = v / vpr;
                    reinterpret_cast<uint4 *>(smap)[yy * (ld / 8) + (v - yy * vpr)] = a[u];
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < npix; i += NT) {
            __half a = o0[i];
            const int y = i / w, x = i - y * w;
            if (flip) a = __hmul(__hadd(a, o1[y * w + (w - 1 - x)]), __float2half(0.5f));
            smap[y * ld + x] = a;
        }
    }
}

template <int NT = kThreads>
__device__ void load_channel(float *smap, const float *o0, const float *o1, int h, int w, bool flip, int ld = 0) {
    if (ld == 0) ld = w;
    const int npix = h * w;
    const bool vec_ok = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(o0) & 15) == 0) &&
                        (!flip || (reinterpret_cast<uintptr_t>(o1) & 15) == 0);
    if (vec_ok) {
        const int nvec = npix / 4;
        const int vpr = w / 4;
        const float4 *p0 = reinterpret_cast<const float4 *>(o0);
        const float4 *p1 = reinterpret_cast<const float4 *>(o1);
        for (int v0 = threadIdx.x; v0 < nvec; v0 += NT * kLoadUnroll) {
            float4 a[kLoadUnroll], m[kLoadUnroll];
#pragma unroll
            for (int u = 0; u < kLoadUnroll; u++) {
                const int v = v0 + u * NT;
                if (v < nvec) {
                    a[u] = p0[v];
                    if (flip) {
                        const int y = v / vpr, vx = v - y * vpr;
                        m[u] = p1[y * vpr + (vpr - 1 - vx)];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kLoadUnroll; u++) {
                const int v = v0 + u * NT;
                if (v < nvec) {
                    if (flip) {
                        a[u].x = __fadd_rn(a[u].x, m[u].w) / 2.0f;
                        a[u].y = __fadd_rn(a[u].y, m[u].z) / 2.0f;
                        a[u].z = __fadd_rn(a[u].z, m[u].y) / 2.0f;
                        a[u].w = __fadd_rn(a[u].w, m[u].x) / 2.0f;
                    }
                    const int yy = v / vpr;
                    reinterpret_cast<float4 *>(smap)[yy * (ld / 4) + (v - yy * vpr)] = a[u];
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < npix; i += NT) {
            float a = o0[i];
            const int y = i / w, x = i - y * w;
            if (flip) a = __fadd_rn(a, o1[y * w + (w - 1 - x)]) / 2.0f;
            smap[y * ld + x] = a;
        }
    }
}

// x4 bicubic (OpenCV INTER_CUBIC restatement, see oracle/posepaf_oracle.c) of an LDS-resident map, evaluated at
// ONE output pixel (X, Y) of the (4*ph, 4*pw) upsample of the window [y0, y0+ph) x [x0, x0+pw) of the map;
// taps clamp to the WINDOW (border replicate).  Horizontal pass on four rows, then vertical, products and sums
// rounded separately, left to right.
template <typename T>
__device__ __forceinline__ float bicubic4_at(const T *smap, int w, int x0, int y0, int pw, int ph, int X, int Y,
                                             const float *s_cub) {
    const int sx = ((X + 2) >> 2) - 1, sy = ((Y + 2) >> 2) - 1;
    const float4 ca = reinterpret_cast<const float4 *>(s_cub)[X & 3];
    const float4 cb = reinterpret_cast<const float4 *>(s_cub)[Y & 3];
    // (an unclamped "interior" fast path was tried: the extra divergent branch made the kernel 1.2-1.8x slower)
    const int xi0 = x0 + clampi(sx - 1, 0, pw - 1), xi1 = x0 + clampi(sx, 0, pw - 1);
    const int xi2 = x0 + clampi(sx + 1, 0, pw - 1), xi3 = x0 + clampi(sx + 2, 0, pw - 1);
    float hrow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const T *row = smap + (y0 + clampi(sy - 1 + j, 0, ph - 1)) * w;
        float v = __fmul_rn(ldsf(row, xi0), ca.x);
        v = __fadd_rn(v, __fmul_rn(ldsf(row, xi1), ca.y));
        v = __fadd_rn(v, __fmul_rn(ldsf(row, xi2), ca.z));
        v = __fadd_rn(v, __fmul_rn(ldsf(row, xi3), ca.w));
        hrow[j] = v;
    }
    float v = __fmul_rn(hrow[0], cb.x);
    v = __fadd_rn(v, __fmul_rn(hrow[1], cb.y));
    v = __fadd_rn(v, __fmul_rn(hrow[2], cb.z));
    v = __fadd_rn(v, __fmul_rn(hrow[3], cb.w));
    return v;
}

// ------------------------------------------------------------------------------------------------ K_A
// LDS layout (dynamic): [map: h*w T][cubic 16 f32][peak linear index i32 x maxp][peak-mask bytes x ceil(h*w/8)]
template <typename T>
__global__ __launch_bounds__(kThreads) void k_heat_peaks(const T *__restrict__ net, int n_samples, int h, int w,
                                                         int flip, int refine, int nms_mode, float thr, int maxp,
                                                         float4 *__restrict__ peaks, int *counts,
                                                         unsigned *__restrict__ status, int *order, int *arrive_all) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int part = blockIdx.x, img = blockIdx.y;
    const int npix = h * w;
    size_t off = 0;
    T *smap = reinterpret_cast<T *>(lds_raw);
    off += (sizeof(T) * (size_t)npix + 15) & ~(size_t)15;
    float *s_cub = reinterpret_cast<float *>(lds_raw + off);
    off += 64;
    int *s_pk = reinterpret_cast<int *>(lds_raw + off);
    off += (4 * (size_t)maxp + 15) & ~(size_t)15;
    unsigned char *s_m8 = lds_raw + off;  // one peak-mask byte per 8-pixel vector
    __shared__ int s_wsum[kWaves];
    __shared__ float s_hpass[kWaves][5 * 20];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 16) s_cub[threadIdx.x] = d_cubic4[threadIdx.x >> 2][threadIdx.x & 3];

    const size_t plane = (size_t)npix;
    const T *o0 = net + ((size_t)img * n_samples * PP_NUM_CH + PP_NUM_LIMB + part) * plane;
    const T *o1 = net + (((size_t)img * n_samples + 1) * PP_NUM_CH + PP_NUM_LIMB + d_flip_heat_ord[part]) * plane;
    long long *stamps = d_stamps;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    stamp(stamps, wg, 0);
    load_channel(smap, o0, o1, h, w, flip != 0);
    __syncthreads();
    stamp(stamps, wg, 1);

    // ---- A3: local maxima.  Each lane tests 8 consecutive row-major pixels per step (one 16-byte LDS read for
    // binary16 maps); only pixels above the threshold (a few per cent) go on to the neighbour reads.  Peak order must
    // be np.nonzero's (ascending linear index): per-(step, wave) counts -> block prefix -> lane prefix -> bit rank.
    const int nvec = (npix + 7) >> 3;
    const int nk = (nvec + kThreads - 1) / kThreads;  // steps; <= kMaxSteps is checked on the host
    const bool rows_aligned = (w & 7) == 0;  // then an 8-pixel vector never straddles two rows
    auto mask8 = [&](int v) -> unsigned {
        const int i0 = v << 3;
        if (i0 >= npix) return 0u;
        float val[8];
        if (i0 + 8 <= npix) {
            load8(smap + i0, val);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) val[j] = i0 + j < npix ? ldsf(smap, i0 + j) : -INFINITY;
        }
        unsigned above = 0;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (nms_mode == 0 ? (val[j] > thr) : (val[j] >= thr)) above |= 1u << j;  // parse_skeletons.py:116 / util.py:184
        if (above == 0) return 0u;
        unsigned m = 0;
        if (rows_aligned) {
            // branch-free form: the rows above/below as two more 16-byte reads, the horizontal neighbours from the
            // vector itself plus one scalar on each side; out-of-map neighbours are -inf (never greater)
            const int y = i0 / w, x0 = i0 - y * w;
            float up[8], dn[8];
            if (y > 0) load8(smap + i0 - w, up);
            if (y < h - 1) load8(smap + i0 + w, dn);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (y == 0) up[j] = -INFINITY;
                if (y == h - 1) dn[j] = -INFINITY;
            }
            const float lft = x0 > 0 ? ldsf(smap, i0 - 1) : -INFINITY;
            const float rgt = x0 + 8 < w ? ldsf(smap, i0 + 8) : -INFINITY;
            float ul = -INFINITY, ur = -INFINITY, dl = -INFINITY, dr = -INFINITY;
            if (nms_mode != 0) {
                if (y > 0 && x0 > 0) ul = ldsf(smap, i0 - w - 1);
                if (y > 0 && x0 + 8 < w) ur = ldsf(smap, i0 - w + 8);
                if (y < h - 1 && x0 > 0) dl = ldsf(smap, i0 + w - 1);
                if (y < h - 1 && x0 + 8 < w) dr = ldsf(smap, i0 + w + 8);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float vj = val[j];
                bool pk = !(up[j] > vj) && !(dn[j] > vj);
                pk = pk && !((j > 0 ? val[j - 1] : lft) > vj) && !((j < 7 ? val[j + 1] : rgt) > vj);
                if (nms_mode != 0) {  // full 3x3 window (utils/util.py:181-184)
                    pk = pk && !((j > 0 ? up[j - 1] : ul) > vj) && !((j < 7 ? up[j + 1] : ur) > vj);
                    pk = pk && !((j > 0 ? dn[j - 1] : dl) > vj) && !((j < 7 ? dn[j + 1] : dr) > vj);
                }
                if (pk) m |= 1u << j;
            }
            return m & above;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (above & (1u << j)) {
                const float vj = val[j];
                const int i = i0 + j;
                const int y = i / w, x = i - y * w;
                bool pk = true;
                if (y > 0 && ldsf(smap, i - w) > vj) pk = false;
                if (y < h - 1 && ldsf(smap, i + w) > vj) pk = false;
                if (x > 0 && ldsf(smap, i - 1) > vj) pk = false;
                if (x < w - 1 && ldsf(smap, i + 1) > vj) pk = false;
                if (nms_mode != 0) {
                    if (y > 0 && x > 0 && ldsf(smap, i - w - 1) > vj) pk = false;
                    if (y > 0 && x < w - 1 && ldsf(smap, i - w + 1) > vj) pk = false;
                    if (y < h - 1 && x > 0 && ldsf(smap, i + w - 1) > vj) pk = false;
                    if (y < h - 1 && x < w - 1 && ldsf(smap, i + w + 1) > vj) pk = false;
                }
                if (pk) m |= 1u << j;
            }
        }
        return m;
    };
    for (int k = 0; k < nk; k++) {
        const int v = k * kThreads + threadIdx.x;
        const unsigned m8 = mask8(v);
        if (v < nvec) s_m8[v] = (unsigned char)m8;
    }
    __syncthreads();
    // Thread t now owns mask bytes [t*bpt, (t+1)*bpt), i.e. a CONTIGUOUS pixel range, so peak order (np.nonzero:
    // ascending linear index) is thread order: one block scan of the per-thread counts gives every peak's rank.
    const int bpt = nk;  // == ceil(nvec / kThreads)
    const int b0 = threadIdx.x * bpt;
    int cnt = 0;
    unsigned long long word = 0;
    const bool one_word = bpt == 8 && (nvec & 7) == 0;  // the 128 x 128 case: a thread's 64 pixels are one 8-byte LDS read
                                                       // (only when no thread's range is partial: a ragged tail takes the byte loop)
    if (one_word) {
        word = b0 + 8 <= nvec ? *reinterpret_cast<const unsigned long long *>(s_m8 + b0) : 0ull;
        cnt = __popcll(word);
    } else {
        for (int q = b0; q < b0 + bpt && q < nvec; q++) cnt += __popc((unsigned)s_m8[q]);
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(incl, d);
        if (lane >= d) incl += n;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int rank = incl - cnt;
    for (int q = 0; q < wave; q++) rank += s_wsum[q];
    int total = 0;
    for (int q = 0; q < kWaves; q++) total += s_wsum[q];
    const int kept = total < maxp ? total : maxp;
    if (one_word) {
        while (word && rank < maxp) {
            const int bit = __ffsll((long long)word) - 1;
            word &= word - 1;
            s_pk[rank++] = (b0 << 3) + bit;
        }
    } else {
        for (int q = b0; q < b0 + bpt && q < nvec && rank < maxp; q++) {
            unsigned m = s_m8[q];
            while (m && rank < maxp) {
                const int j = __ffs(m) - 1;
                m &= m - 1;
                s_pk[rank++] = (q << 3) + j;
            }
        }
    }
    __syncthreads();
    stamp(stamps, wg, 2);

    // ---- A4: per-peak refinement, one wave per peak
    float4 *out = peaks + ((size_t)img * PP_NUM_PART + part) * maxp;
    for (int p = wave; p < kept; p += kWaves) {
        const int i = s_pk[p];
        const int py = i / w, px = i - py * w;
        float ox, oy, score;
        if (refine == 2) {
            // util.refine_centroid (utils/util.py:188-213), radius 2: border peaks are returned unrefined with the raw
            // score; otherwise offset = sum(box * grid) / sum(box) and score = mean(box).  np.mgrid makes x_grid vary
            // along ROWS, so the reference's "offset_x" is the row centroid; restated as written.  Sums in f64.
            if (py - 2 < 0 || py + 3 > h || px - 2 < 0 || px + 3 > w) {
                ox = (float)px;
                oy = (float)py;
                score = ldsf(smap, i);
            } else {
                double sx = 0.0, sy = 0.0, sv = 0.0;
                if (lane < 25) {
                    const int r = lane / 5, c = lane - r * 5;
                    const double v = (double)ldsf(smap, (py - 2 + r) * w + (px - 2 + c));
                    sx = v * (double)(r - 2);
                    sy = v * (double)(c - 2);
                    sv = v;
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    sx += __shfl_xor(sx, d);
                    sy += __shfl_xor(sy, d);
                    sv += __shfl_xor(sv, d);
                }
                ox = (float)((double)px + sx / sv);
                oy = (float)((double)py + sy / sv);
                score = (float)(sv / 25.0);
            }
        } else if (refine == 3) {
            ox = (float)px;
            oy = (float)py;
            score = ldsf(smap, i);
        } else if (refine == 1) {
            const int x_min = px - 2 < 0 ? 0 : px - 2, y_min = py - 2 < 0 ? 0 : py - 2;  // win_size 2, :135,:143-144
            const int x_max = px + 2 > w - 1 ? w - 1 : px + 2, y_max = py + 2 > h - 1 ? h - 1 : py + 2;
            const int pw = x_max - x_min + 1, ph = y_max - y_min + 1;
            const int uw = pw * 4, n = uw * ph * 4;
            // separable evaluation, same arithmetic as the per-pixel form: the horizontal pass of every patch row
            // is computed once (ph x uw values, kept in this wave's LDS scratch), the vertical pass reads 4 of them
            float *hp = s_hpass[wave];
            for (int k = lane; k < ph * uw; k += 64) {
                const int j = k / uw, col = k - j * uw;
                const int sx = ((col + 2) >> 2) - 1;
                const float4 ca = reinterpret_cast<const float4 *>(s_cub)[col & 3];
                const T *row = smap + (y_min + j) * w + x_min;
                float v = __fmul_rn(ldsf(row, clampi(sx - 1, 0, pw - 1)), ca.x);
                v = __fadd_rn(v, __fmul_rn(ldsf(row, clampi(sx, 0, pw - 1)), ca.y));
                v = __fadd_rn(v, __fmul_rn(ldsf(row, clampi(sx + 1, 0, pw - 1)), ca.z));
                v = __fadd_rn(v, __fmul_rn(ldsf(row, clampi(sx + 2, 0, pw - 1)), ca.w));
                hp[k] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // this wave's LDS writes precede its reads below
            __builtin_amdgcn_wave_barrier();
            float best_v = -INFINITY;
            int best_k = 0x7fffffff;
            for (int k = lane; k < n; k += 64) {
                const int row = k / uw, col = k - row * uw;
                const int sy = ((row + 2) >> 2) - 1;
                const float4 cb = reinterpret_cast<const float4 *>(s_cub)[row & 3];
                float v = __fmul_rn(hp[clampi(sy - 1, 0, ph - 1) * uw + col], cb.x);
                v = __fadd_rn(v, __fmul_rn(hp[clampi(sy, 0, ph - 1) * uw + col], cb.y));
                v = __fadd_rn(v, __fmul_rn(hp[clampi(sy + 1, 0, ph - 1) * uw + col], cb.z));
                v = __fadd_rn(v, __fmul_rn(hp[clampi(sy + 2, 0, ph - 1) * uw + col], cb.w));
                if (v > best_v || best_k == 0x7fffffff) {
                    best_v = v;
                    best_k = k;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // arg-max with first-occurrence tie-break (ndarray.argmax, :156)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const float ov = __shfl_xor(best_v, d);
                const int ok = __shfl_xor(best_k, d);
                if (ok != 0x7fffffff && (best_k == 0x7fffffff || ov > best_v || (ov == best_v && ok < best_k))) {
                    best_v = ov;
                    best_k = ok;
                }
            }
            const int row = best_k / uw, col = best_k - row * uw;
            ox = (float)(4 * x_min + col);  // :164-171 collapses to stride*x_min + col (exact integer)
            oy = (float)(4 * y_min + row);
            score = best_v;
        } else {
            ox = __fadd_rn(__fmul_rn(__fadd_rn((float)px, 0.5f), 4.0f), -0.5f);  // compute_resized_coords, :122-123
            oy = __fadd_rn(__fmul_rn(__fadd_rn((float)py, 0.5f), 4.0f), -0.5f);
            score = ldsf(smap, i);
        }
        if (lane == 0) out[p] = make_float4(ox, oy, score, 0.0f);
    }
    if (threadIdx.x == 0) {
        // write-through (sc1) so that the sorting workgroup below reads this launch's count; correctness never depends on it
        __hip_atomic_store(counts + img * PP_NUM_PART + part, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status[img * kFlagWords + part] = total > maxp ? PP_ST_PEAK_OVERFLOW : 0u;  // plain store, every launch
    }
    __syncthreads();
    stamp(stamps, wg, 3);
    if (!order) return;
    // ---- the LAST workgroup of the grid orders the images by estimated matching load (sum over limbs of nA * nB), heaviest
    // first, for K_B's dispatch.  A stale count can only make the order worse, never wrong: the ranks below always form a
    // permutation of 0..B-1 (ties broken by index) because they are computed from ONE consistent copy in LDS.
    __shared__ int s_sorter;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(arrive_all, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (int)(gridDim.x * gridDim.y) - 1;
        if (last) __hip_atomic_store(arrive_all, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
        s_sorter = last;
    }
    __syncthreads();
    if (!s_sorter) return;
    const int B = gridDim.y;
    int *s_w = reinterpret_cast<int *>(lds_raw);  // the map is no longer needed (the host checked that B ints fit)
    for (int i = threadIdx.x; i < B; i += kThreads) {
        int c[PP_NUM_PART];
#pragma unroll
        for (int p = 0; p < PP_NUM_PART; p++) {
            const int v = __hip_atomic_load(counts + i * PP_NUM_PART + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c[p] = v < maxp ? v : maxp;
        }
        int wsum = 0;
#pragma unroll
        for (int l = 0; l < PP_NUM_LIMB; l++) wsum += c[kLimbA[l]] * c[kLimbB[l]];
        s_w[i] = wsum;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += kThreads) {
        const int wi = s_w[i];
        int rank = 0;
        for (int j = 0; j < B; j++) {
            const int wj = s_w[j];
            rank += (wj > wi) || (wj == wi && j < i);
        }
        order[rank] = i;
    }
}

// ------------------------------------------------------------------------------------------------ K_B
// LDS row stride of K_B's limb map: the row length plus 16 bytes.  Lanes of a wave gather along near-vertical limbs (torso,
// legs): with a dense 128-column map equal columns of ALL rows share a bank and those gathers serialise (44 % of the
// kernel's LDS cycles were bank conflicts, profiles/r02_pmc_postproc_kernels.json); 16 bytes keep the vector stores aligned.
template <typename T>
__host__ __device__ constexpr int limb_map_ld(int w) { return w + 16 / (int)sizeof(T); }

// Samplers.  Both return the value the reference reads at PAF(y, x, limb) (pafprocess.cpp:9, :322).
template <typename T>
struct LdsBicubicSampler {  // fused path: the (4h, 4w) upsample is never materialised
    const T *smap;
    const float *s_cub;
    int h, w, ld;  // ld: LDS row stride in elements (>= w)
    __device__ __forceinline__ float at(int X, int Y) const {
        X = clampi(X, 0, 4 * w - 1);  // the reference does not bounds-check; in-range peaks never leave the map
        Y = clampi(Y, 0, 4 * h - 1);
        return bicubic4_at(smap, ld, 0, 0, w, h, X, Y, s_cub);
    }
};
struct GlobalHwcSampler {  // drop-in path: caller supplies the already up-sampled (H, W, C) map
    const float *paf;
    int H, W, C, limb;
    __device__ __forceinline__ float at(int X, int Y) const {
        X = clampi(X, 0, W - 1);
        Y = clampi(Y, 0, H - 1);
        return paf[((size_t)Y * W + X) * C + limb];
    }
};

constexpr int kSortStack = 48;

// libstdc++ std::sort (introsort, threshold 16, depth limit 2*floor(log2 n), heapsort fallback, final
// insertion sort) with the reference's comparator a.overall_score >= b.overall_score
// (pafprocess.cpp:109, :333-335), executed on LDS arrays (key[i], gen[i]) by one lane (run_partitions) or by one
// wave with lane-parallel partitions (run_partitions_wave, the form K_B uses).  Tied keys are common on
// this path (duplicate peaks) and the order among ties decides the greedy matching, so the algorithm itself is
// part of the result; it is restated step for step (see oracle/posepaf_oracle.c for the same restatement in C).
// It only runs when a limb has MORE than 16 accepted candidates AND at least one exact tie -- otherwise the
// sorted order is unique, or (n <= 16: pure insertion sort) has a closed form, and ranks are computed in parallel.
// Scans that libstdc++ leaves unguarded stop at the array bounds here and set oob.
struct SortElem {
    float key;
    int gen;
};
// element accessor: LDS arrays touched by one lane.  (A variant that kept the arrays in the VGPRs of one wave and
// moved elements with v_readlane was measured 1.4x SLOWER: the extra instructions of a 4-register select cost more
// issue slots on a lone wave than the LDS round trips they replace.)
struct LdsSortAcc {
    float *key;
    int *gen;
    __device__ __forceinline__ SortElem get(int i) const { return SortElem{key[i], gen[i]}; }
    __device__ __forceinline__ void put(int i, const SortElem &e) const {
        key[i] = e.key;
        gen[i] = e.gen;
    }
};
template <typename Acc>
struct StdSortGE {
    Acc A;  // by value so that the register-resident form stays in registers
    int n;
    bool oob;
    __device__ StdSortGE(const Acc &acc, int n_) : A(acc), n(n_), oob(false) {}
    __device__ __forceinline__ static bool ge(const SortElem &a, const SortElem &c) { return a.key >= c.key; }
    __device__ __forceinline__ SortElem get(int i) const { return A.get(i); }
    __device__ __forceinline__ void put(int i, const SortElem &e) { A.put(i, e); }
    __device__ __forceinline__ void swp(int i, int j) {
        const SortElem t = get(i);
        put(i, get(j));
        put(j, t);
    }
    __device__ void push_heap(int first, int hole, int top, SortElem value) {
        int parent = (hole - 1) / 2;
        while (hole > top && ge(get(first + parent), value)) {
            put(first + hole, get(first + parent));
            hole = parent;
            parent = (hole - 1) / 2;
        }
        put(first + hole, value);
    }
    __device__ void adjust_heap(int first, int hole, int len, SortElem value) {
        const int top = hole;
        int child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (ge(get(first + child), get(first + child - 1))) child--;
            put(first + hole, get(first + child));
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            put(first + hole, get(first + child - 1));
            hole = child - 1;
        }
        push_heap(first, hole, top, value);
    }
    __device__ void heapsort(int first, int last) {
        const int len = last - first;
        if (len >= 2) {
            int parent = (len - 2) / 2;
            while (true) {
                adjust_heap(first, parent, len, get(first + parent));
                if (parent == 0) break;
                parent--;
            }
        }
        while (last - first > 1) {
            --last;
            const SortElem v = get(last);
            put(last, get(first));
            adjust_heap(first, 0, last - first, v);
        }
    }
    __device__ void move_median_to_first(int result, int a, int m, int c) {
        const SortElem va = get(a), vm = get(m), vc = get(c);
        if (ge(va, vm)) {
            if (ge(vm, vc)) swp(result, m);
            else if (ge(va, vc)) swp(result, c);
            else swp(result, a);
        } else if (ge(va, vc)) swp(result, a);
        else if (ge(vm, vc)) swp(result, c);
        else swp(result, m);
    }
    // ---- the __introsort_loop part of std::sort (partitions, and the heapsort fallback, down to pieces of <= 16
    // elements), executed by ONE WAVE: all 64 lanes call it with uniform arguments; every partition is done by the whole
    // wave, the median-of-3 and the (rare) heapsort fallback by lane 0.  The explicit stack (3 * kSortStack ints of LDS)
    // replaces the recursion: libstdc++ recurses into the right part and loops on the left one; the parts are disjoint, so
    // the order in which they are processed does not change the result.  With the non-strict comparator a left scan may run
    // past `last` (every element of the range >= pivot) and stop in a neighbouring range; libstdc++ then simply continues with
    // [first, cut) -- reproduced as is (a scan that would leave the array stops at its end and sets oob).
    // __unguarded_partition(first, last, pivot) with the comparator `>=`:
    //   the left scan stops only on elements  < pivot ("left stoppers",  L_1 < L_2 < ... in index order, from `first` up),
    //   the right scan only on elements       > pivot ("right stoppers", R_1 > R_2 > ... from `last - 1` down);
    //   elements equal to the pivot stop neither.  The k-th round swaps L_k with R_k as long as L_k < R_k, so the result is:
    //   swap the first k pairs (k = number of i with L_i < R_i -- monotone, so a count), return min(L_{k+1}, R_k)
    //   (R_k now holds an element < pivot).  Stoppers outside [first, last) can never be swapped (an outside L is right of
    //   every R and vice versa); they only decide where an unguarded scan ends, or that it leaves the array (oob).
    // sL / sR: LDS scratch, one int per element of the range each.
    __device__ int partition_wave(int first, int last, int pivot, int lane, int *sL, int *sR) {
        const float pv = A.key[pivot];
        int nL = 0, nR = 0;
        for (int base = first; base < last; base += 64) {
            const int i = base + lane;
            const bool is = i < last && A.key[i] < pv;
            const unsigned long long m = __ballot(is);
            if (is) sL[nL + __popcll(m & lanemask_lt())] = i;
            nL += __popcll(m);
        }
        for (int base = last - 1; base >= first; base -= 64) {
            const int j = base - lane;
            const bool is = j >= first && A.key[j] > pv;
            const unsigned long long m = __ballot(is);
            if (is) sR[nR + __popcll(m & lanemask_lt())] = j;
            nR += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        const int mn = nL < nR ? nL : nR;
        int k = 0;
        for (int base = 0; base < mn; base += 64) {
            const int i = base + lane;
            k += __popcll(__ballot(i < mn && sL[i] < sR[i]));
        }
        for (int base = 0; base < k; base += 64) {
            const int i = base + lane;
            if (i < k) swp(sL[i], sR[i]);
        }
        __builtin_amdgcn_wave_barrier();
        int cut;
        if (k >= 1) {
            const int rk = sR[k - 1];
            cut = (k < nL && sL[k] < rk) ? sL[k] : rk;
        } else if (nL > 0) {
            cut = sL[0];
        } else {  // no element < pivot in the range: the reference's scan runs on into the following elements
            cut = n;
            for (int base = last; base < n && cut == n; base += 64) {
                const int i = base + lane;
                const unsigned long long m = __ballot(i < n && A.key[i] < pv);
                if (m) cut = base + __ffsll((long long)m) - 1;
            }
            if (cut == n) oob = true;
        }
        if (k == 0 && nR == 0) {  // no element > pivot in the range: the right scan runs down past the pivot
            bool found = false;
            for (int base = first - 1; base >= 0 && !found; base -= 64) {
                const int j = base - lane;
                found = __ballot(j >= 0 && A.key[j] > pv) != 0;
            }
            if (!found) oob = true;
        }
        return cut;
    }
    __device__ void run_partitions_wave(int *stk, int lane, int *sL, int *sR) {
        oob = false;
        if (n <= 0) return;
        int lg = 0;
        while ((1 << (lg + 1)) <= n) lg++;
        int sp = 1;
        stk[0] = 0;  // every lane stores the same values
        stk[1] = n;
        stk[2] = 2 * lg;
        while (sp > 0) {
            --sp;
            const int first = stk[3 * sp];
            int last = stk[3 * sp + 1];
            int depth = stk[3 * sp + 2];
            while (last - first > 16) {
                if (depth == 0) {
                    if (lane == 0) heapsort(first, last);
                    __builtin_amdgcn_wave_barrier();
                    break;
                }
                --depth;
                const int mid = first + (last - first) / 2;
                if (lane == 0) move_median_to_first(first, first + 1, mid, last - 1);
                __builtin_amdgcn_wave_barrier();
                const int cut = partition_wave(first + 1, last, first, lane, sL, sR);
                if (sp < kSortStack) {
                    stk[3 * sp] = cut;
                    stk[3 * sp + 1] = last;
                    stk[3 * sp + 2] = depth;
                    ++sp;
                }
                last = cut;
            }
        }
    }
};

// One (a, b) peak pair: pafprocess.cpp:66-106 + get_paf_scores :311-327, split so that the samples can be taken in two
// instalments (see connect_limb): geometry, samples [i0, i1) accumulated IN INDEX ORDER, final criteria.
struct PairGeom {
    float vec_length, step_x, step_y;
    int num_steps;
};
__device__ __forceinline__ bool pair_geom(int ax, int ay, int bx, int by, PairGeom &g) {
    const float vx = (float)(bx - ax), vy = (float)(by - ay);
    g.vec_length = sqrtf(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)));  // :70
    if ((double)g.vec_length < 1e-12) return false;                          // :71
    int n = (int)((double)__fadd_rn(g.vec_length, 1.0f) + 0.5);              // round2int, :73, :329
    g.num_steps = n > 20 ? 20 : n;                                           // STEP_PAF
    g.step_x = vx / (float)(g.num_steps - 1);                                // :314-315
    g.step_y = vy / (float)(g.num_steps - 1);
    return true;
}
template <typename Sampler>
__device__ __forceinline__ void pair_samples(const Sampler &smp, const PairGeom &g, int ax, int ay, int i0, int i1, float &scores,
                                             int &criterion1) {
    for (int i = i0; i < i1; i++) {
        const int lx = (int)((double)__fadd_rn((float)ax, __fmul_rn((float)i, g.step_x)) + 0.5);  // :318-319
        const int ly = (int)((double)__fadd_rn((float)ay, __fmul_rn((float)i, g.step_y)) + 0.5);
        const float s = smp.at(lx, ly);
        scores = __fadd_rn(scores, s);  // :86
        if (s > 0.1f) criterion1 += 1;  // THRESH_PAF_SCORE
    }
}
// smallest count that passes `criterion1 > num_steps * 0.8f` (:93-95)
__device__ __forceinline__ int pair_min_count(int num_steps) { return (int)floorf(__fmul_rn((float)num_steps, 0.8f)) + 1; }
__device__ __forceinline__ bool pair_finish(const PairGeom &g, float scores, int criterion1, float as, float bs, int min_img_size,
                                            float *c2_out, float *overall_out) {
    double prior = 0.5 * (double)min_img_size / (double)g.vec_length - 1.0;  // :92
    if (!(prior < 0.0)) prior = 0.0;                                          // std::min(0.0, prior)
    const float criterion2 = (float)((double)(scores / (float)g.num_steps) + prior);
    const float min_num_steps = __fmul_rn((float)g.num_steps, 0.8f);  // :93 THRESH_PAF_STEP_RATIO
    if (!((float)criterion1 > min_num_steps && criterion2 > 0.0f)) return false;  // :95
    // :96-98  PAF_OUT_WEIGHTS = {0.5, 0.25, 0.25}
    *overall_out = __fadd_rn(__fadd_rn(__fmul_rn(0.5f, criterion2), __fmul_rn(0.25f, as)), __fmul_rn(0.25f, bs));
    *c2_out = criterion2;
    return true;
}

// K_B working set in LDS (after the map): peaks of the two parts, the accepted candidates (generation order) and
// the matching state.  Bytes: 40*maxp + 28*cap.
struct LimbLds {
    int *ax, *ay, *bx, *by;   // [maxp] Peak.x / Peak.y (ints)
    float *as, *bs;           // [maxp] Peak.score
    int *minA, *minB;         // [maxp] lowest rank among live candidates touching this endpoint
    int *usedA, *usedB;       // [maxp]
    float *key;               // [cap] overall_score
    int *rank;                // [cap] position of candidate g in the reference's sorted order
    int *order;               // [cap] inverse: candidate at sorted position r
    int *state;               // [cap] 0 live, 1 accepted, 2 dead
    float *c_score;           // [cap] criterion2
    float *c_len;             // [cap]
    unsigned *c_idx;          // [cap] ia | ib << 16
};
__host__ __device__ inline size_t limb_lds_bytes(int maxp, int cap) { return 40 * (size_t)maxp + 28 * (size_t)cap; }

__device__ inline LimbLds carve_limb_lds(unsigned char *p, int maxp, int cap) {
    LimbLds L;
    int *q = reinterpret_cast<int *>(p);
    L.ax = q; q += maxp;
    L.ay = q; q += maxp;
    L.bx = q; q += maxp;
    L.by = q; q += maxp;
    L.as = reinterpret_cast<float *>(q); q += maxp;
    L.bs = reinterpret_cast<float *>(q); q += maxp;
    L.minA = q; q += maxp;
    L.minB = q; q += maxp;
    L.usedA = q; q += maxp;
    L.usedB = q; q += maxp;
    L.key = reinterpret_cast<float *>(q); q += cap;
    L.rank = q; q += cap;
    L.order = q; q += cap;
    L.state = q; q += cap;
    L.c_score = reinterpret_cast<float *>(q); q += cap;
    L.c_len = reinterpret_cast<float *>(q); q += cap;
    L.c_idx = reinterpret_cast<unsigned *>(q);
    return L;
}

// pafprocess.cpp:51-130 for one limb, all 256 threads:
//  1. score every (a, b) pair (one lane each), ordered compaction of the accepted ones (generation order)
//  2. rank of every candidate in the reference's sorted order (parallel; exact tie semantics, see StdSortGE)
//  3. greedy matching as repeated acceptance of locally dominant candidates: a live candidate whose rank is the
//     lowest among the live candidates sharing its a-peak AND among those sharing its b-peak is exactly a
//     candidate the sequential scan of :113-129 accepts; endpoints are then retired.  Ranks are distinct, so this
//     yields the same set as the sequential greedy pick (and min(nA, nB) of :111 can never bind earlier).
//  4. accepted connections written in rank order (the order the assembly consumes them in).
template <typename Sampler, int NT = kThreads>
__device__ int connect_limb(const Sampler &smp, const LimbLds &L, int nA, int nB, int cap, int maxp, int min_img_size,
                             float4 *__restrict__ conn_out, int *__restrict__ conn_count,
                             unsigned *__restrict__ status_word, long long *stamps = nullptr, int wg = 0,
                             float4 *__restrict__ aux_out = nullptr, int offA = 0, int offB = 0) {
    __shared__ int s_wcnt[2][(NT / 64)];
    __shared__ int s_stack[3 * kSortStack];
    __shared__ int s_oob;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npairs = nA * nB;
    int ncand = 0;  // uniform across the workgroup
    int buf = 0;
    // Scoring in two instalments.  (1) Every pair takes its first kFirst samples; a pair whose misses (samples <= 0.1)
    // already exceed what `criterion1 > 0.8 n` (:93-95) tolerates can never be accepted and is dropped -- most of the
    // nA x nB pairs join different people and die here.  (2) The survivors of several rounds are packed together (in
    // generation order) and finish their remaining samples with all lanes busy, continuing the SAME running sum, so
    // the float additions happen in the reference's order.  The survivor list lives in the rank/order/state arrays,
    // which are not needed before the ranking step.
    constexpr int kFirst = 8;
    int *sv_pair = L.rank;
    float *sv_sum = reinterpret_cast<float *>(L.order);
    int *sv_cnt = L.state;
    int nsv = 0;  // uniform
    auto flush = [&]() {
        for (int sb = 0; sb < nsv; sb += NT, buf ^= 1) {
            const int sidx = sb + threadIdx.x;
            bool ok = false;
            float c2 = 0.f, overall = 0.f, len = 0.f;
            int ia = 0, ib = 0;
            if (sidx < nsv) {
                const int p = sv_pair[sidx];
                ia = p / nB;
                ib = p - ia * nB;
                PairGeom g;
                pair_geom(L.ax[ia], L.ay[ia], L.bx[ib], L.by[ib], g);
                float scores = sv_sum[sidx];
                int c1 = sv_cnt[sidx];
                if (g.num_steps > kFirst) pair_samples(smp, g, L.ax[ia], L.ay[ia], kFirst, g.num_steps, scores, c1);
                ok = pair_finish(g, scores, c1, L.as[ia], L.bs[ib], min_img_size, &c2, &overall);
                len = g.vec_length;
            }
            const unsigned long long m = __ballot(ok);
            if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int k = 0; k < (NT / 64); k++) {
                const int c = s_wcnt[buf][k];
                if (k < wave) before += c;
                all += c;
            }
            if (ok) {
                const int pos = ncand + before + __popcll(m & lanemask_lt());
                if (pos < cap) {
                    L.key[pos] = overall;
                    L.c_score[pos] = c2;
                    L.c_len[pos] = len;
                    L.c_idx[pos] = (unsigned)ia | ((unsigned)ib << 16);
                }
            }
            ncand += all;
        }
        __syncthreads();  // the survivor arrays may be overwritten from here on
        nsv = 0;
    };
    // Few pairs (most limbs of most images: P people give ~P * P pairs): one lane per pair leaves the wave a single chain of
    // twenty dependent bicubic samples (~8 us for ANY number of pairs <= 256).  Below NT / 2 pairs the SAMPLES are spread over
    // lanes instead -- eight lanes take the eight first samples of a pair, sixteen lanes the remaining twelve of a survivor --
    // and the running sum is then formed by every lane of the group in the reference's order (s0 + s1 + ...; lanes past the
    // pair's step count contribute +0.0f, the identity: a running sum that starts at +0.0f is never -0.0f).  Same values, same
    // additions in the same order, a dependent chain five times shorter.
    const bool sample_parallel = npairs <= d_sp_max_pairs;
    if (sample_parallel) {
        constexpr int G1 = 8, PP1 = NT / G1;     // lanes per pair / pairs per round, first instalment
        constexpr int G2 = 16, PP2 = NT / G2;    // second instalment: samples kFirst .. 19 (twelve of the sixteen lanes work)
        static_assert(kFirst == G1 && 20 - kFirst <= G2, "lane groups cover the two instalments");
        for (int base = 0; base < npairs; base += PP1, buf ^= 1) {
            const int p = base + (int)(threadIdx.x / G1), j = threadIdx.x & (G1 - 1);
            const int gl = lane & ~(G1 - 1);     // first lane of this pair's group
            bool alive = false;
            float scores = 0.0f, sj = 0.0f;
            int k1 = 0, nsteps = 0;
            if (p < npairs) {
                const int ia = p / nB, ib = p - ia * nB;
                PairGeom g;
                if (pair_geom(L.ax[ia], L.ay[ia], L.bx[ib], L.by[ib], g)) {
                    nsteps = g.num_steps;
                    k1 = g.num_steps < kFirst ? g.num_steps : kFirst;
                    if (j < k1) {
                        int c_unused = 0;
                        pair_samples(smp, g, L.ax[ia], L.ay[ia], j, j + 1, sj, c_unused);   // sj = 0 + s_j = s_j exactly
                    }
                }
            }
            const unsigned long long hit = __ballot(j < k1 && sj > 0.1f);
#pragma unroll
            for (int k = 0; k < G1; k++) scores = __fadd_rn(scores, __shfl(sj, gl + k));   // s_0 + s_1 + ... in order
            const int c1 = __popcll((hit >> gl) & ((1ull << G1) - 1ull));
            if (nsteps > 0) alive = (k1 - c1) <= nsteps - pair_min_count(nsteps);
            const unsigned long long m = __ballot(alive && j == 0);
            if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int k = 0; k < (NT / 64); k++) {
                const int c = s_wcnt[buf][k];
                if (k < wave) before += c;
                all += c;
            }
            if (alive && j == 0) {
                const int pos = nsv + before + __popcll(m & lanemask_lt());
                sv_pair[pos] = p;
                sv_sum[pos] = scores;
                sv_cnt[pos] = c1;
            }
            nsv += all;   // <= npairs <= NT / 2 <= cap: never flushed early
        }
        __syncthreads();
        for (int sb = 0; sb < nsv; sb += PP2, buf ^= 1) {
            const int sidx = sb + (int)(threadIdx.x / G2), j = threadIdx.x & (G2 - 1);
            const int gl = lane & ~(G2 - 1);
            bool ok = false;
            float c2 = 0.f, overall = 0.f, len = 0.f, sj = 0.0f, scores = 0.0f;
            int ia = 0, ib = 0, c1 = 0, nsteps = 0;
            PairGeom g;
            if (sidx < nsv) {
                const int p = sv_pair[sidx];
                ia = p / nB;
                ib = p - ia * nB;
                pair_geom(L.ax[ia], L.ay[ia], L.bx[ib], L.by[ib], g);
                nsteps = g.num_steps;
                scores = sv_sum[sidx];
                c1 = sv_cnt[sidx];
                if (kFirst + j < nsteps) {
                    int c_unused = 0;
                    pair_samples(smp, g, L.ax[ia], L.ay[ia], kFirst + j, kFirst + j + 1, sj, c_unused);
                }
            }
            const unsigned long long hit = __ballot(kFirst + j < nsteps && sj > 0.1f);
#pragma unroll
            for (int k = 0; k < 20 - kFirst; k++) scores = __fadd_rn(scores, __shfl(sj, gl + k));
            c1 += __popcll((hit >> gl) & ((1ull << G2) - 1ull));
            if (sidx < nsv && j == 0) {
                ok = pair_finish(g, scores, c1, L.as[ia], L.bs[ib], min_img_size, &c2, &overall);
                len = g.vec_length;
            }
            const unsigned long long m = __ballot(ok);
            if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int k = 0; k < (NT / 64); k++) {
                const int c = s_wcnt[buf][k];
                if (k < wave) before += c;
                all += c;
            }
            if (ok) {
                const int pos = ncand + before + __popcll(m & lanemask_lt());
                if (pos < cap) {
                    L.key[pos] = overall;
                    L.c_score[pos] = c2;
                    L.c_len[pos] = len;
                    L.c_idx[pos] = (unsigned)ia | ((unsigned)ib << 16);
                }
            }
            ncand += all;
        }
        __syncthreads();
        nsv = 0;
    }
    for (int base = 0; !sample_parallel && base < npairs; base += NT, buf ^= 1) {
        if (nsv + NT > cap) {
            __syncthreads();  // the previous round's survivors (written by other waves) must be visible to flush()
            flush();
        }
        const int p = base + threadIdx.x;
        bool alive = false;
        float scores = 0.0f;
        int c1 = 0;
        if (p < npairs) {
            const int ia = p / nB;  // generation order of the reference: a outer, b inner (:61-64)
            const int ib = p - ia * nB;
            PairGeom g;
            if (pair_geom(L.ax[ia], L.ay[ia], L.bx[ib], L.by[ib], g)) {
                const int k1 = g.num_steps < kFirst ? g.num_steps : kFirst;
                pair_samples(smp, g, L.ax[ia], L.ay[ia], 0, k1, scores, c1);
                alive = (k1 - c1) <= g.num_steps - pair_min_count(g.num_steps);
            }
        }
        const unsigned long long m = __ballot(alive);
        if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < (NT / 64); k++) {
            const int c = s_wcnt[buf][k];
            if (k < wave) before += c;
            all += c;
        }
        if (alive) {
            const int pos = nsv + before + __popcll(m & lanemask_lt());
            sv_pair[pos] = p;
            sv_sum[pos] = scores;
            sv_cnt[pos] = c1;
        }
        nsv += all;
    }
    __syncthreads();
    flush();
    unsigned st = 0;
    if (ncand > cap) {
        st |= PP_ST_CAND_OVERFLOW;
        ncand = cap;
    }
    const int n = ncand;
    stamp(stamps, wg, 2);
    if (threadIdx.x == 0) s_oob = 0;
    for (int i = threadIdx.x; i < maxp; i += NT) {
        L.usedA[i] = 0;
        L.usedB[i] = 0;
    }
    __syncthreads();

    // ---- 2. ranks (counting: gt = keys strictly greater, eq = keys equal incl. itself; 4 VALU per comparison pair)
    bool tie = false;
    for (int t = threadIdx.x; t < n; t += NT) {
        const float kt = L.key[t];
        int gt = 0, eq = 0;
#pragma unroll 4
        for (int j = 0; j < n; j++) {
            const float kj = L.key[j];  // broadcast read
            gt += kj > kt;
            eq += kj == kt;
        }
        // n <= 16 is a pure insertion sort with `>=`: an element moves in front of every earlier EQUAL element, so
        // among equals the later-generated comes first.  Without ties the order is unique.  (With ties and n > 16 the
        // ranks come from the emulation below instead.)
        int eq_later = 0;
        if (eq > 1 && n <= 16)
            for (int j = t + 1; j < n; j++) eq_later += L.key[j] == kt;
        L.rank[t] = gt + eq_later;
        L.state[t] = 0;
        tie |= eq > 1;
    }
    const int any_tie = __syncthreads_or(tie);
    if (n > 16 && any_tie) {
        for (int t = threadIdx.x; t < n; t += NT) L.order[t] = t;
        __syncthreads();
        if (wave == 0) {  // one wave; the preliminary ranks and the state array serve as its scratch
            LdsSortAcc acc{L.key, L.order};  // permuted in place: the generation-indexed keys are no longer needed
            StdSortGE<LdsSortAcc> srt(acc, n);
            srt.run_partitions_wave(s_stack, lane, L.rank, L.state);
            if (lane == 0 && srt.oob) s_oob = 1;
        }
        __syncthreads();
        // Final insertion sort in closed form from the post-partition array: position p ends at rank
        // (#keys > key[p]) + (#equal keys after p).  The reference's unguarded insert of an element at p >= 16 runs off
        // the array when nothing strictly greater precedes it, i.e. when key[p] >= max(key[0..p-1]): inclusive prefix
        // maxima by a doubling scan in the two scratch arrays.
        float *pm = reinterpret_cast<float *>(L.state), *pm2 = reinterpret_cast<float *>(L.rank);
        for (int p = threadIdx.x; p < n; p += NT) pm[p] = L.key[p];
        __syncthreads();
        for (int d = 1; d < n; d <<= 1) {
            for (int p = threadIdx.x; p < n; p += NT) {
                const float v = pm[p];
                pm2[p] = p >= d ? fmaxf(v, pm[p - d]) : v;
            }
            __syncthreads();
            float *tmp = pm;
            pm = pm2;
            pm2 = tmp;
        }
        bool oob = false;
        for (int p = 16 + threadIdx.x; p < n; p += NT) oob |= L.key[p] >= pm[p - 1];
        if (__syncthreads_or(oob) && threadIdx.x == 0) s_oob = 1;  // (barrier: the scratch arrays are rewritten below)
        for (int p = threadIdx.x; p < n; p += NT) {
            const float kp = L.key[p];
            int gt = 0, eq = 0;
#pragma unroll 4
            for (int q = 0; q < n; q++) {
                const float kq = L.key[q];  // broadcast read
                gt += kq > kp;
                eq += kq == kp;
            }
            int eq_after = 0;
            if (eq > 1)
                for (int q = p + 1; q < n; q++) eq_after += L.key[q] == kp;
            L.rank[L.order[p]] = gt + eq_after;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += NT) {
            L.order[L.rank[t]] = t;
            L.state[t] = 0;
        }
    } else {
        for (int t = threadIdx.x; t < n; t += NT) L.order[L.rank[t]] = t;
    }
    __syncthreads();
    if (s_oob) st |= PP_ST_SORT_UNDEFINED;

    stamp(stamps, wg, 3);
    // ---- 3. greedy matching by local dominance
    for (int pass = 0; pass <= n; pass++) {  // every pass accepts at least the best live candidate: <= n passes
        for (int i = threadIdx.x; i < maxp; i += NT) {
            L.minA[i] = 0x7fffffff;
            L.minB[i] = 0x7fffffff;
        }
        __syncthreads();
        bool live = false;
        for (int t = threadIdx.x; t < n; t += NT) {
            if (L.state[t] == 0) {
                const unsigned idx = L.c_idx[t];
                const int ia = (int)(idx & 0xffffu), ib = (int)(idx >> 16);
                if (L.usedA[ia] || L.usedB[ib]) {
                    L.state[t] = 2;
                } else {
                    atomicMin(&L.minA[ia], L.rank[t]);
                    atomicMin(&L.minB[ib], L.rank[t]);
                    live = true;
                }
            }
        }
        if (!__syncthreads_or(live)) break;
        for (int t = threadIdx.x; t < n; t += NT) {
            if (L.state[t] == 0) {
                const unsigned idx = L.c_idx[t];
                const int ia = (int)(idx & 0xffffu), ib = (int)(idx >> 16);
                const int r = L.rank[t];
                if (L.minA[ia] == r && L.minB[ib] == r) {
                    L.state[t] = 1;
                    L.usedA[ia] = 1;
                    L.usedB[ib] = 1;
                }
            }
        }
        __syncthreads();
    }

    stamp(stamps, wg, 4);
    // ---- 4. ordered output
    int ncn = 0;
    for (int base = 0; base < n; base += NT, buf ^= 1) {
        const int r = base + threadIdx.x;
        bool acc = false;
        int t = 0;
        if (r < n) {
            t = L.order[r];
            acc = L.state[t] == 1;
        }
        const unsigned long long m = __ballot(acc);
        if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < (NT / 64); k++) {
            const int c = s_wcnt[buf][k];
            if (k < wave) before += c;
            all += c;
        }
        if (acc) {
            const unsigned idx = L.c_idx[t];
            const int ia = (int)(idx & 0xffffu), ib = (int)(idx >> 16);
            const int o = ncn + before + __popcll(m & lanemask_lt());
            // REQUIREMENT of the publication protocol (k_limb_connect): every datum the image's assembly workgroup reads from
            // this workgroup -- these two rows, the count, the status word -- must leave with a write-through (sc1) store; the
            // publisher then only drains its stores (vmcnt(0)), meets at the barrier and stores the flag, with no release fence.
            // A plain store here would sit in this XCD's L2 and the assembly (possibly on another XCD) could read stale bytes.
            store_sc1(conn_out + o, make_float4(__int_as_float(ia), __int_as_float(ib), L.c_score[t], L.c_len[t]));
            // what the assembly needs besides: the two peaks' ids (position in the part-ordered peak line,
            // pafprocess.cpp:34, :43-48) and their scores (pl[id].score, :162, :266)
            if (aux_out) store_sc1(aux_out + o, make_float4(__int_as_float(offA + ia), __int_as_float(offB + ib), L.as[ia], L.bs[ib]));
        }
        ncn += all;
    }
    if (threadIdx.x == 0) {
        store_sc1(conn_count, ncn);
        store_sc1(status_word, st);  // a store on every launch (no memset / atomic-OR protocol)
    }
    stamp(stamps, wg, 5);
    return ncn;   // workgroup-uniform
}

// ------------------------------------------------------------------------------------------------ K_C, wave form
// pafprocess.cpp:132-285 for ONE image by ONE wave -- the form the batched path uses, either as the tail of K_B (the last
// limb workgroup of an image to finish runs it: k_limb_connect) or as a kernel of its own (k_assemble_wave: timing,
// diagnostics).  Same results as k_assemble above, different mechanics:
//   * STABLE SLOTS.  A skeleton is born into the next free slot and never moves; `skeletons.erase` (:228) only marks the
//     slot dead.  The reference's vector order is the birth order of the live skeletons, so "first match" / "second match"
//     of the scan (:143-150) are the lowest / second-lowest matching slot and the output order is slot order.  No row shifts.
//     (Slots run out only after 256 BIRTHS; the table is then compacted, order preserved, and only more than 256 LIVE
//     skeletons raise PP_ST_SKEL_OVERFLOW, as before.)
//   * The table is part-major, sk[p][slot]: lane s reads/writes column s of a row without bank conflicts.  Per limb every
//     lane caches the six words of "its" skeletons that a limb can touch (ids of the limb's two parts, the part-2 score,
//     count, total, longest limb) in registers; updates go to the registers and write through to LDS.
//   * Classification of a limb's connections (<= 64 per pass) with two small tables instead of a scan per connection:
//     connection lanes publish "end point -> connection", skeleton lanes look their two peaks up and count themselves into
//     cnt[connection] (= the reference's num_found, one per matching skeleton) and flag the connections that share a
//     skeleton.  Connections with cnt <= 1 that share nothing are independent of every other connection of the limb
//     (pairwise different end points): found-1 updates are applied by the owning skeleton lane, births by the connection
//     lane, whole runs at once and in connection order; the others (two skeletons = possible merge, a shared skeleton, a
//     peak held twice) go through the reference's scan one by one -- done on the cached registers with one ballot per 64
//     slots, no LDS round trip.  A merge that ADDS two real ids in the limb's columns (the `id > 0` quirk, :200-226) makes
//     an id the tables of this pass never saw: from there the pass continues one by one (hand-built test scene).
//   * Connections arrive as two float4 per connection written by K_B: (rank1, rank2, score, length) and
//     (peak id 1, peak id 2, peak score 1, peak score 2), so no peak table is needed before the records are written.
constexpr int kSlots = 256;
constexpr int kBanks = kSlots / 64;
constexpr int kDeadId = -2;  // ids are >= -1 (and sums of those + 1): -2 never matches and never arises

__host__ __device__ inline size_t assemble_wave_lds_bytes(int maxp) {
    return (size_t)2 * 20 * kSlots * 4 + (size_t)2 * maxp * 4 + 2 * 64 * 4 + 32 * 4;
}

struct AsmWaveLds {
    int *sk_id;    // [20][kSlots] rows 0..17 peak id per part (-1 empty, kDeadId erased), row 19 part count
    float *sk_sc;  // [20][kSlots] rows 0..17 limb score per part, row 18 total score, row 19 longest limb
    int *cb1, *cb2;  // [maxp] tag | connection lane, by end-point rank within the part
    int *cnt;        // [64] matching skeletons per connection lane (+ 256 per shared skeleton)
    int *off;        // [19] flat peak id of the first peak of each part (+ total)
};
__device__ inline AsmWaveLds carve_asm_wave_lds(unsigned char *p, int maxp) {
    AsmWaveLds A;
    A.sk_id = reinterpret_cast<int *>(p);
    A.sk_sc = reinterpret_cast<float *>(A.sk_id + 20 * kSlots);
    A.cb1 = reinterpret_cast<int *>(A.sk_sc + 20 * kSlots);
    A.cb2 = A.cb1 + maxp;
    A.cnt = A.cb2 + maxp;
    A.off = A.cnt + 64;
    return A;
}

// value of lane `l` (wave-UNIFORM index) as a scalar: the results steer uniform control flow
__device__ __forceinline__ int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__device__ __forceinline__ void wave_lds_sync() {
    // one wave: its LDS operations complete in issue order; only the compiler must not move them across this point
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the six cached words of one skeleton slot for the current limb
struct SkelCache {
    int p1, p2, ct;
    float s2, ln, tt;
};

// :152-180 on the cached skeleton (found exactly one); write-through.  Only the part-2 entry can change.
__device__ __forceinline__ void apply_found1(SkelCache &k, const AsmWaveLds &A, int slot, int part2, int id2, float c_score,
                                             float c_len, float ps2) {
    const float len1 = k.ln;
    const int min_len = (int)__fmul_rn(len1, 16.0f);  // :154 int truncation of length * LIMB_LENGTH_RATE
    bool wr = false;
    if (k.p2 == -1 && (float)min_len > c_len) {
        k.ct += 1;
        k.tt = __fadd_rn(k.tt, __fadd_rn(ps2, c_score));
        wr = true;
    } else if ((k.p2 != id2 && k.s2 <= c_score && (float)min_len > c_len) || (k.p2 == id2 && k.s2 <= c_score)) {
        // :163-180 id/score are overwritten BEFORE the subtraction: total = (total - t) + t, t = pl[id2].score + conn.score
        const float t = __fadd_rn(ps2, c_score);
        k.tt = __fadd_rn(__fadd_rn(k.tt, -t), t);
        wr = true;
    }
    if (wr) {
        k.p2 = id2;
        k.s2 = c_score;
        k.ln = len1 < c_len ? c_len : len1;
        A.sk_id[part2 * kSlots + slot] = k.p2;
        A.sk_sc[part2 * kSlots + slot] = k.s2;
        A.sk_id[19 * kSlots + slot] = k.ct;
        A.sk_sc[19 * kSlots + slot] = k.ln;
        A.sk_sc[18 * kSlots + slot] = k.tt;
    }
}

__device__ __forceinline__ void load_cache(SkelCache &k, const AsmWaveLds &A, int slot, int part1, int part2, bool valid) {
    k.p1 = valid ? A.sk_id[part1 * kSlots + slot] : kDeadId;
    k.p2 = valid ? A.sk_id[part2 * kSlots + slot] : kDeadId;
    k.s2 = valid ? A.sk_sc[part2 * kSlots + slot] : 0.0f;
    k.ct = valid ? A.sk_id[19 * kSlots + slot] : 0;
    k.ln = valid ? A.sk_sc[19 * kSlots + slot] : 0.0f;
    k.tt = valid ? A.sk_sc[18 * kSlots + slot] : 0.0f;
}

// State of one image's assembly that outlives a pass (all wave-uniform).
struct AsmState {
    int nb;        // slots used so far
    unsigned st;   // PP_ST_* raised by the assembly itself
    long long seq_conns, seq_cycles, odd_merges, class_cycles;  // diagnostics (stamps only)
};

// One pass: <= 64 connections of one limb against the skeletons in slots [0, S.nb), NB = number of 64-slot banks those
// slots span (compile time: the per-bank register caches and every per-bank step are unrolled NB times, so the common
// case -- fewer than 64 skeletons so far -- runs a quarter of the instructions of the general one).
template <int NB>
__device__ __forceinline__ void assemble_pass(AsmState &S, const AsmWaveLds &A, int lane, int mc, int tag, int part1, int part2,
                                              int off1, int off2, int cnt1, int cnt2, int id1, int id2, float c_score,
                                              float c_len, float ps1, float ps2, long long *stamps) {
    const bool have = lane < mc;
    const long long tc0 = stamps ? (long long)clock64() : 0;
    // ---- 1. cache the words of every skeleton this limb can touch; publish "end point -> connection lane"
    SkelCache K[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) load_cache(K[b], A, b * 64 + lane, part1, part2, b * 64 + lane < S.nb);
    // One or two connections (the common case of a sparse image: thirty limbs of a person or two): the classification below
    // costs more than the reference's scan itself -- every connection goes through the scan (step 3), which IS the reference's
    // rule for any connection, so the result is the same by construction.
    const bool few = mc <= 2;
    int myk[NB];
    bool isnew = false;
    unsigned long long confm = few ? (1ull << mc) - 1ull : 0ull;
    int f_id2[NB];
    float f_cs[NB], f_cl[NB], f_ps2[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) myk[b] = -1, f_id2[b] = 0, f_cs[b] = f_cl[b] = f_ps2[b] = 0.0f;
    if (!few) {
    if (have) {
        A.cb1[id1 - off1] = tag | lane;  // end points are pairwise different within a limb: no write conflicts
        A.cb2[id2 - off2] = tag | lane;
    }
    A.cnt[lane] = 0;
    wave_lds_sync();
    // ---- 2. classification: cnt[c] = number of skeletons matching connection c (:143-150), + 256 when c shares one
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int r1 = K[b].p1 - off1, r2 = K[b].p2 - off2;  // erased / unused slots: negative
        int k1 = -1, k2 = -1;
        if (r1 >= 0 && r1 < cnt1) {
            const int v = A.cb1[r1];
            if ((v & ~0xff) == tag) k1 = v & 0xff;
        }
        if (r2 >= 0 && r2 < cnt2) {
            const int v = A.cb2[r2];
            if ((v & ~0xff) == tag) k2 = v & 0xff;
        }
        const bool two = k1 >= 0 && k2 >= 0 && k1 != k2;  // one skeleton, two connections: their order matters
        if (k1 >= 0) atomicAdd(&A.cnt[k1], two ? 257 : 1);  // one count per matching SKELETON
        if (k2 >= 0 && k2 != k1) atomicAdd(&A.cnt[k2], two ? 257 : 1);
        myk[b] = two ? -1 : (k1 >= 0 ? k1 : k2);
    }
    wave_lds_sync();
    const int cw = have ? A.cnt[lane] : 0;
    const bool conflict = cw >= 2;  // two or more skeletons (possible merge / no action), or a shared skeleton
    isnew = have && cw == 0;
    confm = __ballot(conflict);
    // connection words for the skeleton lanes that apply a found-1 update in a run
    bool any_par = false;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (myk[b] >= 0 && ((confm >> myk[b]) & 1ull)) myk[b] = -1;  // its connection goes one by one
        any_par |= myk[b] >= 0;
    }
    if (__ballot(any_par)) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int src = myk[b] >= 0 ? myk[b] : lane;
            f_id2[b] = __shfl(id2, src);
            f_cs[b] = __shfl(c_score, src);
            f_cl[b] = __shfl(c_len, src);
            f_ps2[b] = __shfl(ps2, src);
        }
    }
    }   // !few
    if (stamps) S.class_cycles += (long long)clock64() - tc0;
    // ---- 3. in connection order: maximal runs of independent connections in one step, the others one by one
    // (a run's found-1 updates cannot be hoisted in front of an earlier flagged connection: its merge may ADD two ids into an
    // id that a later connection of the pass holds, and that connection must then see it)
    int pos = 0;
    while (pos < mc) {
        const unsigned long long rest = confm >> pos;
        int next = rest ? pos + __ffsll((long long)rest) - 1 : mc;
        next = next < mc ? next : mc;
        if (next > pos) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const bool mine = myk[b] >= pos && myk[b] < next;
                if (__ballot(mine)) {   // wave-uniform: a bank without work in this run skips the update code instead of masking it
                    if (mine) apply_found1(K[b], A, b * 64 + lane, part2, f_id2[b], f_cs[b], f_cl[b], f_ps2[b]);
                }
            }
            const bool born = isnew && lane >= pos && lane < next;
            const unsigned long long mnew = __ballot(born);
            if (mnew) {  // :257-273 new skeletons, slots in connection order
                const int slot = S.nb + __popcll(mnew & lanemask_lt());
                if (born && slot < kSlots) {
                    A.sk_id[part1 * kSlots + slot] = id1;
                    A.sk_sc[part1 * kSlots + slot] = c_score;
                    A.sk_id[part2 * kSlots + slot] = id2;
                    A.sk_sc[part2 * kSlots + slot] = c_score;
                    A.sk_id[19 * kSlots + slot] = 2;
                    A.sk_sc[19 * kSlots + slot] = c_len;
                    A.sk_sc[18 * kSlots + slot] = __fadd_rn(__fadd_rn(ps1, ps2), c_score);
                }
                S.nb += __popcll(mnew);
                if (S.nb > kSlots) {
                    S.nb = kSlots;
                    S.st |= PP_ST_SKEL_OVERFLOW;
                }
            }
        }
        if (next < mc) {  // ---- the reference's scan for connection `next`, on the cached registers
            const long long t0 = stamps ? (long long)clock64() : 0;
            const int u_id1 = rl(id1, next), u_id2 = rl(id2, next);
            const float u_cs = rlf(c_score, next), u_cl = rlf(c_len, next);
            int num_found = 0, idx1 = 0, idx2 = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) {  // slots born during this pass cannot match (pairwise different end points)
                const unsigned long long mk = __ballot(K[b].p1 == u_id1 || K[b].p2 == u_id2);
                if (mk) {
                    if (num_found == 0) {
                        idx1 = b * 64 + __ffsll((long long)mk) - 1;
                        const unsigned long long m2 = mk & (mk - 1);
                        if (m2) idx2 = b * 64 + __ffsll((long long)m2) - 1;
                    } else if (num_found == 1) {
                        idx2 = b * 64 + __ffsll((long long)mk) - 1;
                    }
                    num_found += __popcll(mk);
                }
            }
            if (num_found == 1) {
                const float u_ps2 = rlf(ps2, next);
#pragma unroll
                for (int b = 0; b < NB; b++)
                    if ((idx1 >> 6) == b) {   // wave-uniform: the other banks' copies of the update are skipped, not masked
                        if ((idx1 & 63) == lane) apply_found1(K[b], A, idx1, part2, u_id2, u_cs, u_cl, u_ps2);
                    }
            } else if (num_found == 2) {  // :182-256, one part per lane; LDS is current (write-through)
                wave_lds_sync();
                const bool isp = lane < PP_NUM_PART;
                const int a_i1 = isp ? A.sk_id[lane * kSlots + idx1] : -1, a_i2 = isp ? A.sk_id[lane * kSlots + idx2] : -1;
                const float a_f1 = isp ? A.sk_sc[lane * kSlots + idx1] : 0.0f, a_f2 = isp ? A.sk_sc[lane * kSlots + idx2] : 0.0f;
                const bool a1 = a_i1 > 0, a2 = a_i2 > 0;  // :200-201 id 0 counts as unassigned
                if (__ballot(a1 && a2) == 0) {
                    // :203-214 running minima "min = (min == 0) ? v : min(v, min)": a plain minimum unless a value is exactly 0
                    float min1, min2;
                    if (__ballot((a1 && a_f1 == 0.0f) || (a2 && a_f2 == 0.0f))) {
                        min1 = 0.0f;
                        min2 = 0.0f;
#pragma unroll 1
                        for (int kp = 0; kp < PP_NUM_PART; kp++) {
                            const int i1 = rl(a_i1, kp), i2 = rl(a_i2, kp);
                            const float f1 = rlf(a_f1, kp), f2 = rlf(a_f2, kp);
                            if (i1 > 0) min1 = (min1 == 0.0f) ? f1 : (f1 < min1 ? f1 : min1);
                            if (i2 > 0) min2 = (min2 == 0.0f) ? f2 : (f2 < min2 ? f2 : min2);
                        }
                    } else {
                        const float inf = __int_as_float(0x7f800000);
                        min1 = a1 ? a_f1 : inf;
                        min2 = a2 ? a_f2 : inf;
#pragma unroll
                        for (int d = 16; d >= 1; d >>= 1) {
                            min1 = fminf(min1, __shfl_xor(min1, d));
                            min2 = fminf(min2, __shfl_xor(min2, d));
                        }
                        min1 = rlf(min1, 0);
                        min2 = rlf(min2, 0);
                        if (min1 == inf) min1 = 0.0f;
                        if (min2 == inf) min2 = 0.0f;
                    }
                    const float len1 = A.sk_sc[19 * kSlots + idx1];
                    const int min_len = (int)__fmul_rn(len1, 16.0f);
                    const float lim = __fmul_rn((min2 < min1 ? min2 : min1), 0.7f);  // :220
                    if (u_cs >= lim || u_cl < (float)min_len) {                       // :221 OR
                        // a limb column where BOTH rows hold an id gets their sum + 1: an id this pass's tables never saw
                        const bool odd = __ballot((lane == part1 || lane == part2) && a_i1 >= 0 && a_i2 >= 0) != 0;
                        const float tot = __fadd_rn(A.sk_sc[18 * kSlots + idx1], __fadd_rn(A.sk_sc[18 * kSlots + idx2], u_cs));
                        const int cnt = A.sk_id[19 * kSlots + idx1] + A.sk_id[19 * kSlots + idx2];
                        wave_lds_sync();
                        if (isp) {
                            A.sk_id[lane * kSlots + idx1] = a_i1 + (a_i2 + 1);
                            A.sk_sc[lane * kSlots + idx1] = __fadd_rn(a_f1, __fadd_rn(a_f2, 1.0f));
                            A.sk_id[lane * kSlots + idx2] = kDeadId;  // skeletons.erase(begin + idx2), :228
                        }
                        if (lane == 0) {
                            A.sk_id[19 * kSlots + idx1] = cnt;
                            A.sk_sc[19 * kSlots + idx1] = len1 < u_cl ? u_cl : len1;
                            A.sk_sc[18 * kSlots + idx1] = tot;
                        }
                        wave_lds_sync();
#pragma unroll
                        for (int b = 0; b < NB; b++) {  // refresh the two owners' caches
                            if ((idx1 >> 6) == b && (idx1 & 63) == lane) load_cache(K[b], A, idx1, part1, part2, true);
                            if ((idx2 >> 6) == b && (idx2 & 63) == lane) K[b].p1 = K[b].p2 = kDeadId;
                        }
                        if (odd) {  // from here on every connection of this pass scans, as the reference does
                            confm |= next + 1 < 64 ? ~0ull << (next + 1) : 0ull;
#pragma unroll
                            for (int b = 0; b < NB; b++) myk[b] = -1;
                            S.odd_merges++;
                        }
                    }
                }
            } else if (num_found == 0) {  // :257-273
                if (S.nb < kSlots) {
                    if (lane == 0) {
                        const float u_ps1 = rlf(ps1, next), u_ps2 = rlf(ps2, next);
                        A.sk_id[part1 * kSlots + S.nb] = u_id1;
                        A.sk_sc[part1 * kSlots + S.nb] = u_cs;
                        A.sk_id[part2 * kSlots + S.nb] = u_id2;
                        A.sk_sc[part2 * kSlots + S.nb] = u_cs;
                        A.sk_id[19 * kSlots + S.nb] = 2;
                        A.sk_sc[19 * kSlots + S.nb] = u_cl;
                        A.sk_sc[18 * kSlots + S.nb] = __fadd_rn(__fadd_rn(u_ps1, u_ps2), u_cs);
                    }
                    S.nb++;
                } else {
                    S.st |= PP_ST_SKEL_OVERFLOW;
                }
            }
            // num_found > 2: no action
            if (stamps) {
                S.seq_cycles += (long long)clock64() - t0;
                S.seq_conns++;
            }
        }
        pos = next + 1;
    }
}

// out of slots: drop the erased ones, order preserved (rare: more than 256 births in one image)
__device__ __forceinline__ void assemble_compact(const AsmWaveLds &A, int lane, int &nb) {
    bool alive[kBanks];
    int newpos[kBanks];
    int base = 0;
#pragma unroll
    for (int b = 0; b < kBanks; b++) {
        const int slot = b * 64 + lane;
        alive[b] = slot < nb && A.sk_id[slot] != kDeadId;
        const unsigned long long mk = __ballot(alive[b]);
        newpos[b] = base + __popcll(mk & lanemask_lt());
        base += __popcll(mk);
    }
#pragma unroll 1
    for (int row = 0; row < 20; row++) {
        int vi[kBanks];
        float vf[kBanks];
#pragma unroll
        for (int b = 0; b < kBanks; b++) {
            vi[b] = A.sk_id[row * kSlots + b * 64 + lane];
            vf[b] = A.sk_sc[row * kSlots + b * 64 + lane];
        }
        wave_lds_sync();
#pragma unroll
        for (int b = 0; b < kBanks; b++)
            if (alive[b]) {
                A.sk_id[row * kSlots + newpos[b]] = vi[b];
                A.sk_sc[row * kSlots + newpos[b]] = vf[b];
            }
        wave_lds_sync();
        if (row < 18)
#pragma unroll
            for (int b = 0; b < kBanks; b++) {
                const int slot = b * 64 + lane;
                if (slot >= base && slot < nb) {
                    A.sk_id[row * kSlots + slot] = -1;
                    A.sk_sc[row * kSlots + slot] = -1.0f;
                }
            }
    }
    wave_lds_sync();
    nb = base;
}

// Executed by all 64 lanes of ONE wave (lane = threadIdx.x & 63); `lds` holds assemble_wave_lds_bytes(maxp) bytes that no
// other wave touches.  conns / aux: [30][maxp] of this image; cc_g: its 30 connection counts; cnt_g: its 18 peak counts.
// STREAM: the limbs of the image are still being matched by other workgroups of the SAME launch.  Limb l may be read once
// ready[l] carries this launch's tag in its upper 24 bits (the lower 8: its connection count); the wave polls with
// agent-scope loads (bounded: PP_ST_SYNC_TIMEOUT), and looks one limb ahead so that a limb that is already published has
// its connections in flight while the previous limb is assembled (see k_limb_connect for the publishing side).
constexpr int kSpinMax = 1 << 21;   // polls of ~1.5 us: seconds of ACTIVE waiting (a co-tenant of the GPU may delay the limbs' dispatch) before giving up
__device__ __forceinline__ unsigned load_flag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool STREAM>
__device__ __forceinline__ void assemble_image_wave(int img, int lane, int maxp, const float4 *__restrict__ pk_g,
                                                    const int *__restrict__ cnt_g, const float4 *__restrict__ conns,
                                                    const float4 *__restrict__ aux, const int *__restrict__ cc_g,
                                                    const unsigned *__restrict__ flags, pp_record *__restrict__ rec,
                                                    unsigned char *lds, long long *stamps, const unsigned *ready = nullptr,
                                                    unsigned want = 0) {
    const AsmWaveLds A = carve_asm_wave_lds(lds, maxp);
    stamp(stamps, img, 0);
    // ---- counts -> flat id offsets (pafprocess.cpp:43-48 flattens in part order); per-limb connection counts
    int pc = lane < PP_NUM_PART ? cnt_g[lane] : 0;
    int cc = (!STREAM && lane < PP_NUM_LIMB) ? cc_g[lane] : 0;
    // STREAM: `snap` = a snapshot of the image's 30 flag words (lane l: limb l), one load; taken here (in flight while the
    // tables are initialised) and again whenever a prefetch is issued
    unsigned snap = (STREAM && lane < PP_NUM_LIMB) ? load_flag(ready + lane) : 0u;
    pc = pc < maxp ? pc : maxp;
    cc = cc < maxp ? cc : maxp;
    int inc = pc, cinc = cc;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const int t = __shfl_up(inc, d), ct = __shfl_up(cinc, d);
        if (lane >= d) {
            inc += t;
            cinc += ct;
        }
    }
    const int offv = inc - pc;  // lane p: flat id of part p's first peak
    // the limb table in a register (lane l: the two parts of limb l): a per-limb memory load would also wait for the
    // prefetched connections (s_waitcnt vmcnt counts every vector memory operation of the wave)
    const int lpv = lane < PP_NUM_LIMB ? ((int)d_limb_pairs[lane][0] | ((int)d_limb_pairs[lane][1] << 8)) : 0;
    const int n_peaks = rl(inc, PP_NUM_PART - 1);
    int n_conn = rl(cinc, PP_NUM_LIMB - 1);   // STREAM: summed up as the limbs arrive
    if (lane <= PP_NUM_PART) A.off[lane] = lane < PP_NUM_PART ? offv : n_peaks;
    // first limb's connections on their way while the table is initialised
    const int m0 = STREAM ? 0 : rl(cc, 0);
    float4 pre_cn = make_float4(0.f, 0.f, 0.f, 0.f), pre_ax = pre_cn;
    if (lane < m0) {
        pre_cn = conns[lane];
        pre_ax = aux[lane];
    }
    // STREAM: connections travel in UNITS -- one limb in the 64-lane layout (lane = connection), or up to eight consecutive
    // published limbs with at most eight connections each in the wide layout (lane = 8 * limb + connection): a sparse image
    // (most images) then costs four round trips to memory instead of thirty.  `cur` = the unit being assembled (blk_cn / blk_ax),
    // `pf` = the unit in flight (pre_cn / pre_ax), chosen from the snapshot taken when the previous unit was issued.
    int cur_first = 0, cur_end = 0, pf_first = 0, pf_n = 0;
    bool cur_wide = false, pf_wide = false, sync_dead = false;
    unsigned cur_fl = 0, pf_fl = 0;
    float4 blk_cn = pre_cn, blk_ax = pre_cn;
    auto issue_unit = [&](int start, unsigned fl) {   // fl: a snapshot; leaves pf_n = 0 when limb `start` is not published in it
        pf_n = 0;
        if (start >= PP_NUM_LIMB || sync_dead) return;
        const unsigned long long pub = __ballot(lane < PP_NUM_LIMB && (fl >> 8) == want);
        const unsigned long long sml = __ballot(lane < PP_NUM_LIMB && (fl & 0xffu) <= 8u);
        if (!((pub >> start) & 1ull)) return;
        int run = __builtin_ctzll(~((pub & sml) >> start));   // consecutive published limbs with <= 8 connections from `start`
        run = run > 8 ? 8 : run;
        pf_first = start;
        pf_fl = fl;
        if (run >= 2) {
            pf_wide = true;
            pf_n = run;
            const int g = lane >> 3, idx = lane & 7;
            const int l = start + (g < run ? g : 0);
            const int ml = (int)(__shfl((int)fl, l) & 0xff);
            if (g < run && idx < ml) {
                pre_cn = conns[(size_t)l * maxp + idx];
                pre_ax = aux[(size_t)l * maxp + idx];
            }
        } else {
            pf_wide = false;
            pf_n = 1;
            const int ml = rl((int)fl, start) & 0xff;
            if (lane < ml) {
                pre_cn = conns[(size_t)start * maxp + lane];
                pre_ax = aux[(size_t)start * maxp + lane];
            }
        }
    };
    {  // every slot starts as an empty skeleton: ids -1, scores -1 (rows 18 / 19 are set at birth); lookup tables: tag 0
        int4 *qi = reinterpret_cast<int4 *>(A.sk_id);
        float4 *qf = reinterpret_cast<float4 *>(A.sk_sc);
        const int4 mi = make_int4(-1, -1, -1, -1);
        const float4 mf = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
        for (int i = lane; i < 18 * kSlots / 4; i += 64) {
            qi[i] = mi;
            qf[i] = mf;
        }
        for (int i = lane; i < 2 * maxp; i += 64) A.cb1[i] = 0;
    }
    wave_lds_sync();
    stamp(stamps, img, 1);

    AsmState S{0, 0u, 0, 0, 0, 0};
    for (int limb = 0; limb < PP_NUM_LIMB; limb++) {
        int m;
        float4 cur_cn, cur_ax;
        if (STREAM) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unit in flight (if any) and the snapshot have landed
            if (limb >= cur_end) {
                if (pf_n == 0 && !sync_dead) {   // nothing on its way: look until this limb is published, fetch, wait (the latency is
                                                 // exposed only when the assembly has caught up with the matching)
                    int spins = 0;
                    while ((((unsigned)rl((int)snap, limb)) >> 8) != want) {
                        if (++spins > kSpinMax) {
                            sync_dead = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        snap = lane < PP_NUM_LIMB ? load_flag(ready + lane) : 0u;
                    }
                    issue_unit(limb, snap);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (sync_dead) {   // give up: the remaining limbs count as empty, the record carries PP_ST_SYNC_TIMEOUT
                    cur_first = limb, cur_end = PP_NUM_LIMB, cur_wide = false, cur_fl = 0u;
                } else {
                    blk_cn = pre_cn, blk_ax = pre_ax;
                    asm volatile("" : "+v"(blk_cn.x), "+v"(blk_cn.y), "+v"(blk_cn.z), "+v"(blk_cn.w), "+v"(blk_ax.x), "+v"(blk_ax.y),
                                 "+v"(blk_ax.z), "+v"(blk_ax.w));
                    cur_first = pf_first, cur_end = pf_first + pf_n, cur_wide = pf_wide, cur_fl = pf_fl;
                    issue_unit(cur_end, snap);   // the next unit (from the snapshot of one unit ago) flies while this one is assembled
                    if (cur_end < PP_NUM_LIMB) snap = lane < PP_NUM_LIMB ? load_flag(ready + lane) : 0u;
                }
            }
            m = rl((int)cur_fl, limb) & 0xff;
            n_conn += m;
            if (cur_wide) {   // this limb's connections sit in lanes 8 * (limb - cur_first) ...: bring them to lanes 0 ..
                const int src = ((limb - cur_first) * 8 + lane) & 63;
                cur_cn = make_float4(__shfl(blk_cn.x, src), __shfl(blk_cn.y, src), __shfl(blk_cn.z, src), __shfl(blk_cn.w, src));
                cur_ax = make_float4(__shfl(blk_ax.x, src), __shfl(blk_ax.y, src), __shfl(blk_ax.z, src), __shfl(blk_ax.w, src));
            } else {
                cur_cn = blk_cn, cur_ax = blk_ax;
            }
        } else {
            m = rl(cc, limb);
            // this limb's connections were requested one limb ago: wait for them HERE, before the next request goes out, so
            // that no later wait of this iteration has to cover the new request as well (vmcnt retires in order)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            cur_cn = pre_cn, cur_ax = pre_ax;
            asm volatile("" : "+v"(cur_cn.z), "+v"(cur_cn.w), "+v"(cur_ax.x), "+v"(cur_ax.y), "+v"(cur_ax.z), "+v"(cur_ax.w));
            if (limb + 1 < PP_NUM_LIMB) {  // next limb's first 64 connections: in flight during this limb
                const int mn = rl(cc, limb + 1);
                if (lane < mn) {
                    pre_cn = conns[(size_t)(limb + 1) * maxp + lane];
                    pre_ax = aux[(size_t)(limb + 1) * maxp + lane];
                }
            }
        }
        if (m == 0) continue;
        const int lp2 = rl(lpv, limb);
        const int part1 = lp2 & 0xff, part2 = lp2 >> 8;
        const int off1 = rl(offv, part1), off2 = rl(offv, part2);
        const int cnt1 = rl(pc, part1), cnt2 = rl(pc, part2);
        {  // first pass: the prefetched connections (the only pass unless max_peaks_per_part > 64)
            const int mc = m < 64 ? m : 64;
            if (S.nb + mc > kSlots) assemble_compact(A, lane, S.nb);
            const int id1 = __float_as_int(cur_ax.x), id2 = __float_as_int(cur_ax.y);
            const int tag = (2 * limb + 1) << 8;  // unique per pass; the tables never need re-zeroing
            if (S.nb <= 64)
                assemble_pass<1>(S, A, lane, mc, tag, part1, part2, off1, off2, cnt1, cnt2, id1, id2, cur_cn.z, cur_cn.w, cur_ax.z,
                                 cur_ax.w, stamps);
            else if (S.nb <= 128)
                assemble_pass<2>(S, A, lane, mc, tag, part1, part2, off1, off2, cnt1, cnt2, id1, id2, cur_cn.z, cur_cn.w, cur_ax.z,
                                 cur_ax.w, stamps);
            else
                assemble_pass<kBanks>(S, A, lane, mc, tag, part1, part2, off1, off2, cnt1, cnt2, id1, id2, cur_cn.z, cur_cn.w,
                                      cur_ax.z, cur_ax.w, stamps);
        }
        if (m > 64) {  // connections 64.. of a limb (its own loads and its own copy of the pass: a load here would otherwise make
                       // every wait of the common path cover the prefetch as well)
            const int mc = m - 64;
            float4 cn2 = make_float4(0.f, 0.f, 0.f, 0.f), ax2 = cn2;
            if (lane < mc) {
                cn2 = conns[(size_t)limb * maxp + 64 + lane];
                ax2 = aux[(size_t)limb * maxp + 64 + lane];
            }
            if (S.nb + mc > kSlots) assemble_compact(A, lane, S.nb);
            assemble_pass<kBanks>(S, A, lane, mc, (2 * limb + 2) << 8, part1, part2, off1, off2, cnt1, cnt2, __float_as_int(ax2.x),
                                  __float_as_int(ax2.y), cn2.z, cn2.w, ax2.z, ax2.w, stamps);
        }
    }
    wave_lds_sync();
    stamp(stamps, img, 2);
    if (stamps && lane == 0) {
        stamps[(size_t)img * 8 + 4] = S.seq_conns;
        stamps[(size_t)img * 8 + 5] = S.seq_cycles;
        stamps[(size_t)img * 8 + 6] = S.odd_merges;
        stamps[(size_t)img * 8 + 7] = S.class_cycles;
    }

    // ---- prune (:278-282): survivors in slot (= birth) order; their slots go to a list (the lookup tables' space is free now)

    int n_out = 0;
#pragma unroll
    for (int b = 0; b < kBanks; b++) {
        if (b * 64 >= S.nb) break;
        const int s = b * 64 + lane;
        bool keep = false;
        if (s < S.nb && A.sk_id[s] != kDeadId) {
            const int count = A.sk_id[19 * kSlots + s];
            const float total = A.sk_sc[18 * kSlots + s];
            keep = !(count < 2 || total / (float)count < 0.45f);
        }
        const unsigned long long mk = __ballot(keep);
        if (keep) {
            const int r = n_out + __popcll(mk & lanemask_lt());
            if (r < PP_MAX_HUMANS) A.sk_id[18 * kSlots + r] = s;  // row 18 of the id table is unused: the kept-slot list
        }
        n_out += __popcll(mk);
    }

    wave_lds_sync();
    const int n_rec = n_out < PP_MAX_HUMANS ? n_out : PP_MAX_HUMANS;
    // ---- records: one (human, part) item per lane step -- id from LDS, the peak's x / y / score from the peak table by ID
    for (int i = lane; i < n_rec * PP_NUM_PART; i += 64) {
        const int r = i / PP_NUM_PART, kp = i - r * PP_NUM_PART;
        const int s = A.sk_id[18 * kSlots + r];
        const int id = A.sk_id[kp * kSlots + s];
        int x = 0, y = 0;
        float sc = 0.0f;
        if (id >= 0 && id < n_peaks) {  // the getters index the flattened peak line BY ID (pafprocess.cpp:299-309)
            int part = kp;
            if (id < A.off[kp] || id >= A.off[kp + 1]) {  // an id-sum of a merge: some other part's peak
                part = 0;
                while (id >= A.off[part + 1]) part++;
            }
            const float4 p = pk_g[(size_t)part * maxp + (id - A.off[part])];
            x = (int)p.x;
            y = (int)p.y;
            sc = p.z;
        }
        pp_human *hm = rec->humans + r;
        hm->peak_id[kp] = id;
        hm->x[kp] = x;
        hm->y[kp] = y;
        hm->part_score[kp] = sc;
    }
    for (int r = lane; r < n_rec; r += 64) {
        const int s = A.sk_id[18 * kSlots + r];
        const int count = A.sk_id[19 * kSlots + s];
        rec->humans[r].score = A.sk_sc[18 * kSlots + s] / (float)count;  // get_score, :295-297
        rec->humans[r].n_parts = count;
    }
    const unsigned fl = or_flags(flags, img, 0, lane);
    if (lane == 0) {
        unsigned st = S.st;
        if (n_out > PP_MAX_HUMANS) st |= PP_ST_HUMAN_OVERFLOW;
        if (sync_dead) st |= PP_ST_SYNC_TIMEOUT;
        rec->n_humans = n_rec;
        rec->n_peaks = n_peaks;
        rec->n_connections = n_conn;
        rec->status = fl | st;
    }
    stamp(stamps, img, 3);
}

__global__ __launch_bounds__(64, 3) void k_assemble_wave(int maxp, const float4 *__restrict__ peaks, const int *__restrict__ counts,
                                                      const float4 *__restrict__ conns, const float4 *__restrict__ aux,
                                                      const int *__restrict__ conn_counts, const unsigned *__restrict__ flags,
                                                      pp_record *__restrict__ records) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int img = blockIdx.x;
    assemble_image_wave<false>(img, threadIdx.x, maxp, peaks + (size_t)img * PP_NUM_PART * maxp, counts + img * PP_NUM_PART,
                               conns + (size_t)img * PP_NUM_LIMB * maxp, aux + (size_t)img * PP_NUM_LIMB * maxp,
                               conn_counts + img * PP_NUM_LIMB, flags, records + img, lds_raw, d_stamps);
}

// K_B: LDS layout (dynamic): [map h*w T][cubic 16 f32][LimbLds: 40*maxp + 28*cap bytes]; the assembly tail re-uses the
// whole region (assemble_wave_lds_bytes).
// Grid (30, B): workgroup (limb, position) works on image order[position] -- K_A's last workgroup sorts the images by load,
// heaviest first, so the crowded images' limbs are dispatched first and their assembly overlaps the rest of the batch.
// "Last block done": every limb workgroup of an image publishes its connections (write-through stores drained by every wave,
// workgroup barrier, one relaxed agent-scope ticket on arrive[img]); the workgroup that draws the last of
// the 30 tickets re-arms the counter for the next launch, acquires, and its wave 0 assembles the image while the other
// waves leave.  No workgroup ever waits for another one, so dispatch order and residency cannot deadlock it.
template <typename T, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 3 : 4) void k_limb_connect(const T *__restrict__ net, int n_samples, int h, int w,
                                                           int flip, int maxp, int cap, int min_img_size,
                                                           const int *__restrict__ min_img_size_dev,
                                                           const float4 *__restrict__ peaks,
                                                           const int *__restrict__ counts, float4 *conns, float4 *aux,
                                                           int *conn_counts, unsigned *status, const int *__restrict__ order,
                                                           int *arrive, unsigned *ready, pp_record *records) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ int s_poff[PP_NUM_PART];
    const int limb = blockIdx.x, img = order ? order[blockIdx.y] : blockIdx.y;
    // Fused form (arrive != NULL): grid (31, B).  Workgroups 0..29 of an image match one limb each and PUBLISH it; workgroup 30
    // is the image's ASSEMBLY: one wave that consumes limb 0, 1, ... as each is published (the assembly needs them in this
    // order anyway, pafprocess.cpp:133), so it runs under the matching instead of after it.  arrive[img] is the image's
    // launch counter (touched by this kernel only): every workgroup of the image reads it at its start, limb l publishes
    // ready[img][l] = (counter + 1) << 8 | connection count, the assembly stores counter + 1 when it is done -- no flag is
    // ever reset and a flag of an earlier launch can never be mistaken for this launch's.  The assembly workgroup has the
    // HIGHEST index of its image: a workgroup is dispatched after every workgroup with a lower index of its XCD's queue, and
    // limb workgroups never wait, so everything it waits for is running or done (its polling is bounded nevertheless).
    const unsigned want = arrive ? (((unsigned)arrive[img] + 1u) & 0xffffffu) : 0u;
    if (limb == PP_NUM_LIMB) {
        if (threadIdx.x >= 64) return;
        __builtin_amdgcn_s_setprio(3);   // a lone latency-bound instruction stream next to streaming waves
        assemble_image_wave<true>(img, threadIdx.x, maxp, peaks + (size_t)img * PP_NUM_PART * maxp, counts + img * PP_NUM_PART,
                                  conns + (size_t)img * PP_NUM_LIMB * maxp, aux + (size_t)img * PP_NUM_LIMB * maxp,
                                  conn_counts + img * PP_NUM_LIMB, status, records + img, lds_raw,
                                  d_stamps ? d_stamps + (size_t)gridDim.x * gridDim.y * 8 : nullptr,  // diagnostics: after the limbs'
                                  ready + (size_t)img * PP_NUM_LIMB, want);
        if (threadIdx.x == 0) store_sc1(arrive + img, (int)want);
        return;
    }
    const int pa = d_limb_pairs[limb][0], pb = d_limb_pairs[limb][1];
    int nA = counts[img * PP_NUM_PART + pa], nB = counts[img * PP_NUM_PART + pb];
    nA = nA < maxp ? nA : maxp;
    nB = nB < maxp ? nB : maxp;
    int *cc = conn_counts + img * PP_NUM_LIMB + limb;
    long long *stamps = d_stamps;
    int ncn_out = 0;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (nA == 0 || nB == 0) {  // no candidate pairs: no connections (pafprocess.cpp:56-58, :111)
        if (threadIdx.x == 0) {
            store_sc1(cc, 0);
            store_sc1(status + img * kFlagWords + PP_NUM_PART + limb, 0u);
        }
    } else {
        const int ld = limb_map_ld<T>(w);   // padded LDS rows (see load_channel)
        size_t off = 0;
        T *smap = reinterpret_cast<T *>(lds_raw);
        off += (sizeof(T) * (size_t)h * ld + 15) & ~(size_t)15;
        float *s_cub = reinterpret_cast<float *>(lds_raw + off);
        off += 64;
        LimbLds L = carve_limb_lds(lds_raw + off, maxp, cap);

        if (threadIdx.x < 16) s_cub[threadIdx.x] = d_cubic4[threadIdx.x >> 2][threadIdx.x & 3];
        if (threadIdx.x < 64) {  // flat peak id of each part's first peak (pafprocess.cpp:43-48)
            int c = threadIdx.x < PP_NUM_PART ? counts[img * PP_NUM_PART + threadIdx.x] : 0;
            c = c < maxp ? c : maxp;
            int inc = c;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const int t = __shfl_up(inc, d);
                if ((int)threadIdx.x >= d) inc += t;
            }
            if (threadIdx.x < PP_NUM_PART) s_poff[threadIdx.x] = inc - c;
        }
        const float4 *pka = peaks + ((size_t)img * PP_NUM_PART + pa) * maxp;
        const float4 *pkb = peaks + ((size_t)img * PP_NUM_PART + pb) * maxp;
        for (int i = threadIdx.x; i < nA; i += NT) {
            const float4 p = pka[i];
            L.ax[i] = (int)p.x;  // Peak.x/y are ints: truncation (pafprocess.cpp:35-36)
            L.ay[i] = (int)p.y;
            L.as[i] = p.z;
        }
        for (int i = threadIdx.x; i < nB; i += NT) {
            const float4 p = pkb[i];
            L.bx[i] = (int)p.x;
            L.by[i] = (int)p.y;
            L.bs[i] = p.z;
        }
        const size_t plane = (size_t)h * w;
        const T *o0 = net + ((size_t)img * n_samples * PP_NUM_CH + limb) * plane;
        const T *o1 = net + (((size_t)img * n_samples + 1) * PP_NUM_CH + d_flip_paf_ord[limb]) * plane;
        stamp(stamps, wg, 0);
        load_channel<NT>(smap, o0, o1, h, w, flip != 0, ld);
        __syncthreads();
        stamp(stamps, wg, 1);

        LdsBicubicSampler<T> smp{smap, s_cub, h, w, ld};
        const int mis = min_img_size_dev ? min_img_size_dev[img] : min_img_size;
        const size_t row = ((size_t)img * PP_NUM_LIMB + limb) * maxp;
        ncn_out = connect_limb<LdsBicubicSampler<T>, NT>(smp, L, nA, nB, cap, maxp, mis, conns + row, cc, status + img * kFlagWords + PP_NUM_PART + limb, stamps, wg,
                     aux + row, s_poff[pa], s_poff[pb]);
    }
    if (!arrive) return;  // two-kernel form (timing / diagnostics): k_assemble_wave follows as its own launch

    // ---- publish this limb.  Everything the assembly reads from this workgroup was stored write-through (store_sc1), so there
    // is no release fence: EVERY storing wave drains its stores, the workgroup meets at a barrier, one lane stores the flag.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) store_sc1(ready + (size_t)img * PP_NUM_LIMB + limb, (want << 8) | (unsigned)ncn_out);
}

// Drop-in path: the caller's (H, W, C) up-sampled map lives in global memory (uploaded by process_paf)
__global__ __launch_bounds__(kThreads) void k_limb_connect_hwc(const float *__restrict__ paf, int H, int W, int C,
                                                               int maxp, int cap, int min_img_size,
                                                               const float4 *__restrict__ peaks,
                                                               const int *__restrict__ counts,
                                                               float4 *__restrict__ conns, int *__restrict__ conn_counts,
                                                               unsigned *__restrict__ status) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int limb = blockIdx.x;
    const int pa = d_limb_pairs[limb][0], pb = d_limb_pairs[limb][1];
    int nA = counts[pa], nB = counts[pb];
    nA = nA < maxp ? nA : maxp;
    nB = nB < maxp ? nB : maxp;
    int *cc = conn_counts + limb;
    if (nA == 0 || nB == 0 || limb >= C) {
        if (threadIdx.x == 0) {
            *cc = 0;
            status[PP_NUM_PART + limb] = 0u;
        }
        return;
    }
    size_t off = 0;
    LimbLds L = carve_limb_lds(lds_raw + off, maxp, cap);
    const float4 *pka = peaks + (size_t)pa * maxp;
    const float4 *pkb = peaks + (size_t)pb * maxp;
    for (int i = threadIdx.x; i < nA; i += kThreads) {
        const float4 p = pka[i];
        L.ax[i] = (int)p.x;
        L.ay[i] = (int)p.y;
        L.as[i] = p.z;
    }
    for (int i = threadIdx.x; i < nB; i += kThreads) {
        const float4 p = pkb[i];
        L.bx[i] = (int)p.x;
        L.by[i] = (int)p.y;
        L.bs[i] = p.z;
    }
    __syncthreads();
    GlobalHwcSampler smp{paf, H, W, C, limb};
    connect_limb(smp, L, nA, nB, cap, maxp, min_img_size, conns + (size_t)limb * maxp, cc, status + PP_NUM_PART + limb);
}

// ------------------------------------------------------------------------------------------------ K_C
// One wave per image.  Skeleton table in LDS: entry [s][k], k in [0,18) = {peak id, limb score},
// k = 18 = {-, total score}, k = 19 = {part count, longest limb} (pafprocess.h:36-45, pafprocess.cpp:11-12).
// The scan over live skeletons (pafprocess.cpp:143-150) is done by the 64 lanes with ballots; the matched
// skeleton(s) are then updated by lane 0 with the reference's exact statement order.
constexpr int kMaxSkel = 256;
constexpr int kSkelStride = 21;  // 20 entries, odd stride: the per-lane column reads of the scan are conflict-free

__host__ __device__ inline size_t assemble_lds_bytes(int maxp) {
    return (size_t)kMaxSkel * kSkelStride * 8 + (size_t)PP_NUM_PART * maxp * 16 + (size_t)PP_NUM_LIMB * maxp * 16 + 16 * (size_t)maxp;
}

__global__ __launch_bounds__(64) void k_assemble(int maxp, int explicit_ids, const float4 *__restrict__ peaks,
                                                 const int *__restrict__ counts, const float4 *__restrict__ conns,
                                                 const int *__restrict__ conn_counts, const unsigned *__restrict__ status,
                                                 int flag_first, pp_record *__restrict__ records) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int ntab = PP_NUM_PART * maxp;
    int *sk_id = reinterpret_cast<int *>(lds_raw);                          // [kMaxSkel][21]
    float *sk_sc = reinterpret_cast<float *>(sk_id + kMaxSkel * kSkelStride);
    int *line_x = reinterpret_cast<int *>(sk_sc + kMaxSkel * kSkelStride);  // peak_infos_line, bucket order
    int *line_y = line_x + ntab;
    float *line_s = reinterpret_cast<float *>(line_y + ntab);
    int *ids = reinterpret_cast<int *>(line_s + ntab);                      // [18][maxp] peak id of (part, rank)
    float4 *s_conn = reinterpret_cast<float4 *>(ids + ntab);                // all connections, limb-major, compact
    int *s_own1 = reinterpret_cast<int *>(s_conn + PP_NUM_LIMB * maxp);     // [maxp] skeleton holding peak r at part 1 / 2
    int *s_own2 = s_own1 + maxp;
    int *s_cb1 = s_own2 + maxp;                                             // [maxp] connection using peak r as end point 1 / 2
    int *s_cb2 = s_cb1 + maxp;
    __shared__ int s_off[PP_NUM_PART + 1];
    __shared__ int s_cnt[PP_NUM_PART];
    __shared__ int s_coff[PP_NUM_LIMB + 1];
    __shared__ int s_conf[64];  // per connection of the current limb: shares a skeleton with another connection

    const int *cnt_g = counts + img * PP_NUM_PART;
    const float4 *pk_g = peaks + (size_t)img * PP_NUM_PART * maxp;
    long long *stamps = d_stamps;
    long long seq_cycles = 0, seq_limbs = 0, odd_merges = 0;
    stamp(stamps, img, 0);
    {  // bucket offsets: one count per lane, wave prefix sums (no serial chain of global loads)
        int c = lane < PP_NUM_PART ? cnt_g[lane] : 0;
        int cc = lane < PP_NUM_LIMB ? conn_counts[img * PP_NUM_LIMB + lane] : 0;
        c = c < maxp ? c : maxp;
        cc = cc < maxp ? cc : maxp;
        int inc = c, cinc = cc;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const int t = __shfl_up(inc, d), ct = __shfl_up(cinc, d);
            if (lane >= d) {
                inc += t;
                cinc += ct;
            }
        }
        if (lane < PP_NUM_PART) {
            s_cnt[lane] = c;
            s_off[lane] = inc - c;
            if (lane == PP_NUM_PART - 1) s_off[PP_NUM_PART] = inc;
        }
        if (lane < PP_NUM_LIMB) {
            s_coff[lane] = cinc - cc;
            if (lane == PP_NUM_LIMB - 1) s_coff[PP_NUM_LIMB] = cinc;
        }
    }
    __syncthreads();
    const int n_peaks = s_off[PP_NUM_PART];
    {  // pafprocess.cpp:43-48 flatten in part order: flat index -> (part, rank); loads of different parts overlap
        int part = 0;
        for (int i = lane; i < n_peaks; i += 64) {
            while (i >= s_off[part + 1]) part++;
            const int r = i - s_off[part];
            const float4 p = pk_g[(size_t)part * maxp + r];
            line_x[i] = (int)p.x;
            line_y[i] = (int)p.y;
            line_s[i] = p.z;
            ids[part * maxp + r] = explicit_ids ? __float_as_int(p.w) : i;  // :34 ids follow input order
        }
    }
    __syncthreads();
    {  // every connection of the image into LDS in one flat sweep, (part, rank) end points translated to peak ids
        const int n_conn = s_coff[PP_NUM_LIMB];
        int limb = 0;
        for (int i = lane; i < n_conn; i += 64) {
            while (i >= s_coff[limb + 1]) limb++;
            float4 cn = conns[((size_t)img * PP_NUM_LIMB + limb) * maxp + (i - s_coff[limb])];
            cn.x = __int_as_float(ids[d_limb_pairs[limb][0] * maxp + __float_as_int(cn.x)]);
            cn.y = __int_as_float(ids[d_limb_pairs[limb][1] * maxp + __float_as_int(cn.y)]);
            s_conn[i] = cn;
        }
    }
    __syncthreads();

    for (int i = lane; i < 4 * maxp; i += 64) s_own1[i] = 0;  // own1, own2, cb1, cb2 (contiguous); tag 0 = never valid
    __syncthreads();
    stamp(stamps, img, 1);
    int nskel = 0;  // uniform
    unsigned st = 0;
    // The connections of ONE limb type have pairwise different end points, so they touch pairwise different skeletons
    // unless (a) a skeleton holds the part-1 peak of one connection and the part-2 peak of another, (b) two skeletons
    // hold the same peak, or (c) a connection joins two skeletons (possible merge + erase).  When none of that
    // happens -- checked exactly, per limb -- the sequential scan of pafprocess.cpp:138-275 gives the same result in
    // any order, and the limb's connections are applied in parallel (one lane each): owner tables peak -> skeleton
    // replace the O(#skeletons) scan.  Otherwise the limb is processed connection by connection as the reference does.
    auto sequential_conn = [&](int ci, int part1, int part2) -> int {  // returns the erased skeleton's index or -1
        int erased = -1;
        const float4 cn = s_conn[ci];
        const int id1 = __float_as_int(cn.x), id2 = __float_as_int(cn.y);
        const float c_score = cn.z, c_len = cn.w;
        // pl[id].score with the reference's indexing BY ID into the bucket-ordered line (:162)
        const float ps1 = (id1 >= 0 && id1 < n_peaks) ? line_s[id1] : 0.0f;
        const float ps2 = (id2 >= 0 && id2 < n_peaks) ? line_s[id2] : 0.0f;
        // ---- :143-150 scan all live skeletons
        int num_found = 0, idx1 = 0, idx2 = 0;
        for (int base = 0; base < nskel; base += 64) {
            const int s = base + lane;
            bool hit = false;
            if (s < nskel) hit = (sk_id[s * kSkelStride + part1] == id1) || (sk_id[s * kSkelStride + part2] == id2);
            unsigned long long m = __ballot(hit);
            if (m) {
                if (num_found == 0) {
                    idx1 = base + __ffsll((long long)m) - 1;
                    const unsigned long long m2 = m & (m - 1);
                    if (m2) idx2 = base + __ffsll((long long)m2) - 1;
                } else if (num_found == 1) {
                    idx2 = base + __ffsll((long long)m) - 1;
                }
                num_found += __popcll(m);
            }
        }
        if (num_found == 1) {  // :152-180
            if (lane == 0) {
                int *i1 = sk_id + idx1 * kSkelStride;
                float *f1 = sk_sc + idx1 * kSkelStride;
                const float len1 = f1[19];
                const int min_len = (int)__fmul_rn(len1, 16.0f);  // :154 int truncation of length*LIMB_LENGTH_RATE
                const int cur_id = i1[part2];
                const float cur_sc = f1[part2];
                if (cur_id == -1 && (float)min_len > c_len) {
                    i1[part2] = id2;
                    f1[part2] = c_score;
                    i1[19] += 1;
                    f1[19] = len1 < c_len ? c_len : len1;
                    f1[18] = __fadd_rn(f1[18], __fadd_rn(ps2, c_score));
                } else if ((cur_id != id2 && cur_sc <= c_score && (float)min_len > c_len) ||
                           (cur_id == id2 && cur_sc <= c_score)) {
                    // :163-180 the id/score are overwritten BEFORE the subtraction, so -= and += use the same
                    // operands: total = (total - t) + t with t = pl[id2].score + conn.score
                    i1[part2] = id2;
                    f1[part2] = c_score;
                    const float t = __fadd_rn(ps2, c_score);
                    f1[18] = __fadd_rn(__fadd_rn(f1[18], -t), t);
                    f1[19] = len1 < c_len ? c_len : len1;
                }
            }
            __syncthreads();
        } else if (num_found == 2) {  // :182-256, one part per lane
            int *i1 = sk_id + idx1 * kSkelStride, *i2 = sk_id + idx2 * kSkelStride;
            float *f1 = sk_sc + idx1 * kSkelStride, *f2 = sk_sc + idx2 * kSkelStride;
            const bool isp = lane < PP_NUM_PART;
            const int a_i1 = isp ? i1[lane] : -1, a_i2 = isp ? i2[lane] : -1;
            const float a_f1 = isp ? f1[lane] : 0.0f, a_f2 = isp ? f2[lane] : 0.0f;
            const bool a1 = a_i1 > 0, a2 = a_i2 > 0;  // :200-201 id 0 counts as unassigned
            const bool is_member = __ballot(a1 && a2) != 0;
            int merge = 0;
            if (!is_member) {
                // :203-214 running minima "min = (min == 0) ? v : min(v, min)": a plain minimum unless a value is exactly 0
                float min1, min2;
                if (__ballot((a1 && a_f1 == 0.0f) || (a2 && a_f2 == 0.0f))) {
                    min1 = 0.0f;
                    min2 = 0.0f;
                    for (int kp = 0; kp < PP_NUM_PART; kp++) {
                        if (i1[kp] > 0) min1 = (min1 == 0.0f) ? f1[kp] : (f1[kp] < min1 ? f1[kp] : min1);
                        if (i2[kp] > 0) min2 = (min2 == 0.0f) ? f2[kp] : (f2[kp] < min2 ? f2[kp] : min2);
                    }
                } else {
                    const float inf = __int_as_float(0x7f800000);
                    min1 = a1 ? a_f1 : inf;
                    min2 = a2 ? a_f2 : inf;
#pragma unroll
                    for (int d = 16; d >= 1; d >>= 1) {
                        min1 = fminf(min1, __shfl_xor(min1, d));
                        min2 = fminf(min2, __shfl_xor(min2, d));
                    }
                    min1 = __shfl(min1, 0);
                    min2 = __shfl(min2, 0);
                    if (min1 == inf) min1 = 0.0f;
                    if (min2 == inf) min2 = 0.0f;
                }
                const float len1 = f1[19];
                const int min_len = (int)__fmul_rn(len1, 16.0f);
                const float lim = __fmul_rn((min2 < min1 ? min2 : min1), 0.7f);  // :220
                if (c_score >= lim || c_len < (float)min_len) {                   // :221 OR
                    // a column where BOTH rows hold an id gets their sum + 1: an id no lookup table knows (see below)
                    const bool odd = __ballot((lane == part1 || lane == part2) && a_i1 >= 0 && a_i2 >= 0) != 0;
                    merge = odd ? 3 : 1;
                    const float tot = __fadd_rn(f1[18], __fadd_rn(f2[18], c_score));
                    const int cnt = i1[19] + i2[19];
                    if (isp) {
                        i1[lane] = a_i1 + (a_i2 + 1);
                        f1[lane] = __fadd_rn(a_f1, __fadd_rn(a_f2, 1.0f));
                    }
                    if (lane == 0) {
                        i1[19] = cnt;
                        f1[19] = len1 < c_len ? c_len : len1;
                        f1[18] = tot;
                    }
                }
            }
            __syncthreads();
            if (merge) {  // skeletons.erase(begin + idx2) (:228): every later row moves down one slot, 64 words per
                          // step: all lanes read one row ahead, barrier, all lanes write (uniform trip count)
                const int end = (nskel - 1) * kSkelStride;
                for (int base = idx2 * kSkelStride; base < end; base += 64) {
                    const int i = base + lane;
                    const bool ok = i < end;
                    const int v = ok ? sk_id[i + kSkelStride] : 0;
                    const float w = ok ? sk_sc[i + kSkelStride] : 0.0f;
                    __syncthreads();
                    if (ok) {
                        sk_id[i] = v;
                        sk_sc[i] = w;
                    }
                    __syncthreads();
                }
                nskel--;
                erased = idx2 | ((merge & 2) << 15);
            }
            __syncthreads();
        } else if (num_found == 0) {  // :257-273
            if (nskel < kMaxSkel) {
                if (lane < 20) {
                    int idv = -1;
                    float scv = -1.0f;
                    if (lane == part1) { idv = id1; scv = c_score; }
                    if (lane == part2) { idv = id2; scv = c_score; }
                    if (lane == 19) { idv = 2; scv = c_len; }
                    if (lane == 18) scv = __fadd_rn(__fadd_rn(ps1, ps2), c_score);
                    sk_id[nskel * kSkelStride + lane] = idv;
                    sk_sc[nskel * kSkelStride + lane] = scv;
                }
                nskel++;
            } else {
                st |= PP_ST_SKEL_OVERFLOW;
            }
            __syncthreads();
        }
        // num_found > 2: no action
        return erased;
    };
    for (int limb = 0; limb < PP_NUM_LIMB; limb++) {
        const int part1 = d_limb_pairs[limb][0], part2 = d_limb_pairs[limb][1];
        const int c0 = s_coff[limb], m = s_coff[limb + 1] - c0;
        if (m == 0) continue;
        if (explicit_ids || m > 64 || nskel + m > kMaxSkel) {  // plain reference order
            for (int ci = c0; ci < c0 + m; ci++) sequential_conn(ci, part1, part2);
            continue;
        }
        const int off1 = s_off[part1], off2 = s_off[part2], cnt1 = s_cnt[part1], cnt2 = s_cnt[part2];
        // table entries carry the limb as a tag (tables zeroed once per launch), so nothing is re-initialised per limb
        const int tag = (limb + 1) << 8;
        int id1 = 0, id2 = 0;
        float c_score = 0.0f, c_len = 0.0f;
        if (lane < m) {
            const float4 cn = s_conn[c0 + lane];
            id1 = __float_as_int(cn.x);
            id2 = __float_as_int(cn.y);
            c_score = cn.z;
            c_len = cn.w;
            s_cb1[id1 - off1] = tag | lane;  // end points are distinct within a limb: no write conflicts
            s_cb2[id2 - off2] = tag | lane;
        }
        s_conf[lane] = 0;
        for (int sb = 0; sb < nskel; sb += 64) {  // owner tables: which skeleton holds peak r at part 1 / part 2
            const int sidx = sb + lane;
            if (sidx < nskel) {
                const int r1 = sk_id[sidx * kSkelStride + part1] - off1, r2 = sk_id[sidx * kSkelStride + part2] - off2;
                if (r1 >= 0 && r1 < cnt1) s_own1[r1] = tag | sidx;
                if (r2 >= 0 && r2 < cnt2) s_own2[r2] = tag | sidx;
            }
        }
        __syncthreads();
        for (int sb = 0; sb < nskel; sb += 64) {  // skeleton lanes flag the connections that are not alone on a skeleton
            const int sidx = sb + lane;
            if (sidx < nskel) {
                const int r1 = sk_id[sidx * kSkelStride + part1] - off1, r2 = sk_id[sidx * kSkelStride + part2] - off2;
                int k1 = -1, k2 = -1;
                bool multi1 = false, multi2 = false;
                if (r1 >= 0 && r1 < cnt1) {
                    multi1 = s_own1[r1] != (tag | sidx);  // (b): another skeleton wrote the same entry
                    const int v = s_cb1[r1];
                    if ((v & ~0xff) == tag) k1 = v & 0xff;
                }
                if (r2 >= 0 && r2 < cnt2) {
                    multi2 = s_own2[r2] != (tag | sidx);
                    const int v = s_cb2[r2];
                    if ((v & ~0xff) == tag) k2 = v & 0xff;
                }
                const bool two = k1 >= 0 && k2 >= 0 && k1 != k2;  // (a): one skeleton, two different connections
                if (k1 >= 0 && (multi1 || two)) s_conf[k1] = 1;
                if (k2 >= 0 && (multi2 || two)) s_conf[k2] = 1;
            }
        }
        __syncthreads();
        int idx1 = -1;
        bool conflict = false;
        if (lane < m) {
            int o1 = s_own1[id1 - off1], o2 = s_own2[id2 - off2];
            o1 = (o1 & ~0xff) == tag ? (o1 & 0xff) : -1;
            o2 = (o2 & ~0xff) == tag ? (o2 & 0xff) : -1;
            conflict = s_conf[lane] != 0 || (o1 >= 0 && o2 >= 0 && o1 != o2);  // (c) two skeletons: possible merge
            idx1 = o1 >= 0 ? o1 : o2;
        }
        unsigned long long confm = __ballot(conflict);
        const float ps1 = (lane < m && id1 >= 0 && id1 < n_peaks) ? line_s[id1] : 0.0f;
        const float ps2 = (lane < m && id2 >= 0 && id2 < n_peaks) ? line_s[id2] : 0.0f;
        // ---- connections in order: maximal runs of independent ones in one step each, the others one by one
        int pos = 0;
        while (pos < m) {
            const unsigned long long rest = confm >> pos;
            int next = rest ? pos + __ffsll((long long)rest) - 1 : m;
            next = next < m ? next : m;
            if (next > pos) {
                const bool mine = lane >= pos && lane < next;
                const bool is_new = mine && idx1 < 0;
                const unsigned long long mnew = __ballot(is_new);
                if (mine && idx1 >= 0) {  // :152-180 on a skeleton no other connection of this limb touches
                    int *i1 = sk_id + idx1 * kSkelStride;
                    float *f1 = sk_sc + idx1 * kSkelStride;
                    const float len1 = f1[19];
                    const int min_len = (int)__fmul_rn(len1, 16.0f);
                    const int cur_id = i1[part2];
                    const float cur_sc = f1[part2];
                    if (cur_id == -1 && (float)min_len > c_len) {
                        i1[part2] = id2;
                        f1[part2] = c_score;
                        i1[19] += 1;
                        f1[19] = len1 < c_len ? c_len : len1;
                        f1[18] = __fadd_rn(f1[18], __fadd_rn(ps2, c_score));
                    } else if ((cur_id != id2 && cur_sc <= c_score && (float)min_len > c_len) ||
                               (cur_id == id2 && cur_sc <= c_score)) {
                        i1[part2] = id2;
                        f1[part2] = c_score;
                        const float t = __fadd_rn(ps2, c_score);
                        f1[18] = __fadd_rn(__fadd_rn(f1[18], -t), t);
                        f1[19] = len1 < c_len ? c_len : len1;
                    }
                } else if (is_new) {  // :257-273 new skeleton; slots in connection order
                    const int slot = nskel + __popcll(mnew & lanemask_lt());
                    int *i1 = sk_id + slot * kSkelStride;
                    float *f1 = sk_sc + slot * kSkelStride;
#pragma unroll
                    for (int k = 0; k < 20; k++) {
                        i1[k] = -1;
                        f1[k] = -1.0f;
                    }
                    i1[part1] = id1;
                    f1[part1] = c_score;
                    i1[part2] = id2;
                    f1[part2] = c_score;
                    i1[19] = 2;
                    f1[19] = c_len;
                    f1[18] = __fadd_rn(__fadd_rn(ps1, ps2), c_score);
                }
                nskel += __popcll(mnew);
                __syncthreads();
            }
            if (next < m) {
                const long long t0 = stamps ? (long long)clock64() : 0;
                const int erased = sequential_conn(c0 + next, part1, part2);
                if (erased >= 0) {
                    if (idx1 > (erased & 0xffff)) idx1--;  // skeletons.erase shifted every later skeleton down one slot
                    // a merge that ADDS two real ids in the part-1/part-2 column makes an id no table knows: from
                    // here on every connection of this limb scans the skeletons as the reference does
                    if (erased >> 16) {
                        confm |= ~0ull << next;
                        odd_merges++;
                    }
                }
                if (stamps) {
                    seq_cycles += (long long)clock64() - t0;
                    seq_limbs++;
                }
            }
            pos = next + 1;
        }
    }
    __syncthreads();
    stamp(stamps, img, 2);
    if (stamps && lane == 0) {
        stamps[(size_t)img * 8 + 4] = seq_limbs;
        stamps[(size_t)img * 8 + 5] = seq_cycles;
        stamps[(size_t)img * 8 + 6] = odd_merges;
    }

    // ---- prune (:278-282) + records; order of survivors preserved
    pp_record *rec = records + img;
    int n_out = 0;
    for (int base = 0; base < nskel; base += 64) {
        const int s = base + lane;
        bool keep = false;
        if (s < nskel) {
            const int count = sk_id[s * kSkelStride + 19];
            const float total = sk_sc[s * kSkelStride + 18];
            keep = !(count < 2 || total / (float)count < 0.45f);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int r = n_out + __popcll(m & lanemask_lt());
            if (r < PP_MAX_HUMANS) {
                pp_human *hm = rec->humans + r;
                const int count = sk_id[s * kSkelStride + 19];
                for (int kp = 0; kp < PP_NUM_PART; kp++) {
                    const int id = sk_id[s * kSkelStride + kp];
                    hm->peak_id[kp] = id;
                    const bool ok = id >= 0 && id < n_peaks;
                    hm->x[kp] = ok ? line_x[id] : 0;
                    hm->y[kp] = ok ? line_y[id] : 0;
                    hm->part_score[kp] = ok ? line_s[id] : 0.0f;
                }
                hm->score = sk_sc[s * kSkelStride + 18] / (float)count;  // get_score, :295-297
                hm->n_parts = count;
            }
        }
        n_out += __popcll(m);
    }
    const unsigned fl = or_flags(status, img, flag_first, lane);
    if (lane == 0) {
        if (n_out > PP_MAX_HUMANS) {
            st |= PP_ST_HUMAN_OVERFLOW;
            n_out = PP_MAX_HUMANS;
        }
        rec->n_humans = n_out;
        rec->n_peaks = n_peaks;
        rec->n_connections = s_coff[PP_NUM_LIMB];
        rec->status = fl | st;
    }
    stamp(stamps, img, 3);
}

// ================================================================================================ A8: Python twins
// find_connections (utils/parse_skeletons.py:324-410) and find_humans (:413-600): the rules evaluate.py follows WITHOUT
// --run_cpp.  Same inputs as the C++ path (peaks of K_A, the x4 bicubic limb maps evaluated on the fly), different
// arithmetic: float64 throughout, np.round(np.linspace) sample positions (round-half-even), NumPy's float32 pairwise
// mean, a STABLE descending sort, `>= 0` membership, AND merge condition, np.maximum merge, old-value subtraction.
// Restated with NumPy-2 scalar promotion (see oracle/posepaf_oracle.c, orc_py_find_humans).

struct LimbLdsPy {
    double *ax, *ay, *as, *bx, *by, *bs; // [maxp] peak x, y (not truncated), score -- float64 like the Python lists
    int *minA, *minB, *usedA, *usedB;    // [maxp]
    double *key, *c_score, *c_len;       // [cap] overall, connect_score, limb_len
    int *rank, *order, *state;           // [cap]
    unsigned *c_idx;                     // [cap]
};
__host__ __device__ inline size_t limb_lds_bytes_py(int maxp, int cap) { return 64 * (size_t)maxp + 8 + 40 * (size_t)cap; }

__device__ inline LimbLdsPy carve_limb_lds_py(unsigned char *p, int maxp, int cap) {
    LimbLdsPy L;
    double *f = reinterpret_cast<double *>(p);
    L.ax = f; f += maxp;
    L.ay = f; f += maxp;
    L.as = f; f += maxp;
    L.bx = f; f += maxp;
    L.by = f; f += maxp;
    L.bs = f; f += maxp;
    int *q = reinterpret_cast<int *>(f);
    L.minA = q; q += maxp;
    L.minB = q; q += maxp;
    L.usedA = q; q += maxp;
    L.usedB = q; q += maxp;
    uintptr_t u = (reinterpret_cast<uintptr_t>(q) + 7) & ~(uintptr_t)7;
    double *d = reinterpret_cast<double *>(u);
    L.key = d; d += cap;
    L.c_score = d; d += cap;
    L.c_len = d; d += cap;
    q = reinterpret_cast<int *>(d);
    L.rank = q; q += cap;
    L.order = q; q += cap;
    L.state = q; q += cap;
    L.c_idx = reinterpret_cast<unsigned *>(q);
    return L;
}

// one (src, dst) pair, utils/parse_skeletons.py:344-388.  VT = dtype of the limb map the reference indexes: float32 on the
// refactored path (cv2.resize output), float64 on the original path (predict's accumulators).
__device__ __forceinline__ float vadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double vadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ bool above_thre2(float v) { return v > 0.1f; }   // float32 array > python float 0.1
__device__ __forceinline__ bool above_thre2(double v) { return v > 0.1; }

template <typename Sampler>
__device__ bool score_pair_py(const Sampler &smp, double ax, double ay, double as_, double bx, double by, double bs_,
                              int img_height, double *score_out, double *overall_out, double *len_out) {
    typedef decltype(smp.at(0, 0)) VT;
    const double dx = bx - ax, dy = by - ay;
    const double limb_len = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));  // :352
    if (limb_len == 0.0) return false;                                                // :356
    const long long rn = __double2ll_rn(limb_len + 1.0);                              // round(): half to even
    const int mid_num = rn < 20 ? (int)rn : 20;                                       // :353
    const double div = (double)(mid_num - 1);
    const double stepx = mid_num > 1 ? dx / div : 0.0, stepy = mid_num > 1 ? dy / div : 0.0;
    VT resp[20];
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < 20; t++) {
        VT v = 0;
        if (t < mid_num) {
            // np.linspace: arange(num) * step + start, last element overwritten with stop (:361-362); step == 0 -> start
            double lx = __dadd_rn(__dmul_rn((double)t, stepx), ax), ly = __dadd_rn(__dmul_rn((double)t, stepy), ay);
            if (mid_num > 1 && t == mid_num - 1) {
                lx = bx;
                ly = by;
            }
            v = smp.at((int)__double2ll_rn(lx), (int)__double2ll_rn(ly));  // np.round: half to even
            if (above_thre2(v)) cnt++;                                      // thre2 (:375)
        }
        resp[t] = v;
    }
    // limb_response.mean(): NumPy's pairwise sum in the array's dtype (8 running sums once n >= 8), then / n
    VT sum;
    if (mid_num < 8) {
        sum = 0;
#pragma unroll
        for (int t = 0; t < 7; t++)
            if (t < mid_num) sum = vadd(sum, resp[t]);
    } else {
        VT r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = resp[k];
        const int full = mid_num - (mid_num & 7);
#pragma unroll
        for (int t = 8; t < 16; t++)
            if (t < full) r[t & 7] = vadd(r[t & 7], resp[t]);
        sum = vadd(vadd(vadd(r[0], r[1]), vadd(r[2], r[3])), vadd(vadd(r[4], r[5]), vadd(r[6], r[7])));
#pragma unroll
        for (int t = 8; t < 20; t++)
            if (t >= full && t < mid_num) sum = vadd(sum, resp[t]);
    }
    const VT mean = sum / (VT)mid_num;
    const double prior = 0.5 * (double)img_height / limb_len - 1.0;  // :366
    double connect_score, half_cs;
    if (sizeof(VT) == 4 && 0.0 < prior) {  // python min(prior, 0) -> int 0: float32 + 0 stays float32
        connect_score = (double)mean;
        half_cs = (double)__fmul_rn(0.5f, (float)mean);
    } else {                                // float64 arithmetic (float32 + float64, or a float64 map)
        connect_score = 0.0 < prior ? (double)mean : __dadd_rn((double)mean, prior);
        half_cs = __dmul_rn(0.5, connect_score);
    }
    if (!((double)cnt > __dmul_rn((double)mid_num, 0.8) && connect_score > 0.0)) return false;  // :375-378
    *overall_out = __dadd_rn(__dadd_rn(half_cs, __dmul_rn(0.25, as_)), __dmul_rn(0.25, bs_));  // :381
    *score_out = connect_score;
    *len_out = limb_len;
    return true;
}

// pafprocess-free Python rules for one limb: scoring, stable ranking, greedy pick, ordered output
template <typename Sampler>
__device__ void connect_limb_py(const Sampler &smp, const LimbLdsPy &L, int nA, int nB, int cap, int maxp, int ih,
                                double4 *__restrict__ conn_out, int *__restrict__ cc, unsigned *__restrict__ status_word) {
    __shared__ int s_wcnt[2][kWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- scoring + ordered compaction (generation order: src outer, dst inner)
    const int npairs = nA * nB;
    int ncand = 0, buf = 0;
    for (int base = 0; base < npairs; base += kThreads, buf ^= 1) {
        const int p = base + threadIdx.x;
        bool ok = false;
        double sc = 0, ov = 0, ln = 0;
        int ia = 0, ib = 0;
        if (p < npairs) {
            ia = p / nB;
            ib = p - ia * nB;
            ok = score_pair_py(smp, L.ax[ia], L.ay[ia], L.as[ia], L.bx[ib], L.by[ib], L.bs[ib], ih, &sc, &ov, &ln);
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < kWaves; k++) {
            const int c = s_wcnt[buf][k];
            if (k < wave) before += c;
            all += c;
        }
        if (ok) {
            const int pos = ncand + before + __popcll(m & lanemask_lt());
            if (pos < cap) {
                L.key[pos] = ov;
                L.c_score[pos] = sc;
                L.c_len[pos] = ln;
                L.c_idx[pos] = (unsigned)ia | ((unsigned)ib << 16);
            }
        }
        ncand += all;
    }
    unsigned st = 0;
    if (ncand > cap) {
        st |= PP_ST_CAND_OVERFLOW;
        ncand = cap;
    }
    const int n = ncand;
    __syncthreads();
    // ---- sorted(reverse=True) is stable: equal keys keep generation order (:391)
    for (int t = threadIdx.x; t < n; t += kThreads) {
        const double kt = L.key[t];
        int r = 0;
        for (int j = 0; j < n; j++) {
            const double kj = L.key[j];
            r += (kj > kt) || (kj == kt && j < t);
        }
        L.rank[t] = r;
        L.state[t] = 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += kThreads) L.order[L.rank[t]] = t;
    __syncthreads();
    // ---- greedy pick (:393-407) as repeated acceptance of locally dominant candidates (see connect_limb)
    for (int pass = 0; pass <= n; pass++) {  // every pass accepts at least the best live candidate: <= n passes
        for (int i = threadIdx.x; i < maxp; i += kThreads) {
            L.minA[i] = 0x7fffffff;
            L.minB[i] = 0x7fffffff;
        }
        __syncthreads();
        bool live = false;
        for (int t = threadIdx.x; t < n; t += kThreads) {
            if (L.state[t] == 0) {
                const unsigned idx = L.c_idx[t];
                const int ia = (int)(idx & 0xffffu), ib = (int)(idx >> 16);
                if (L.usedA[ia] || L.usedB[ib]) {
                    L.state[t] = 2;
                } else {
                    atomicMin(&L.minA[ia], L.rank[t]);
                    atomicMin(&L.minB[ib], L.rank[t]);
                    live = true;
                }
            }
        }
        if (!__syncthreads_or(live)) break;
        for (int t = threadIdx.x; t < n; t += kThreads) {
            if (L.state[t] == 0) {
                const unsigned idx = L.c_idx[t];
                const int ia = (int)(idx & 0xffffu), ib = (int)(idx >> 16);
                const int r = L.rank[t];
                if (L.minA[ia] == r && L.minB[ib] == r) {
                    L.state[t] = 1;
                    L.usedA[ia] = 1;
                    L.usedB[ib] = 1;
                }
            }
        }
        __syncthreads();
    }
    int ncn = 0;
    for (int base = 0; base < n; base += kThreads, buf ^= 1) {
        const int r = base + threadIdx.x;
        bool acc = false;
        int t = 0;
        if (r < n) {
            t = L.order[r];
            acc = L.state[t] == 1;
        }
        const unsigned long long m = __ballot(acc);
        if (lane == 0) s_wcnt[buf][wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < kWaves; k++) {
            const int c = s_wcnt[buf][k];
            if (k < wave) before += c;
            all += c;
        }
        if (acc) {
            const unsigned idx = L.c_idx[t];
            conn_out[ncn + before + __popcll(m & lanemask_lt())] =
                make_double4((double)(idx & 0xffffu), (double)(idx >> 16), L.c_score[t], L.c_len[t]);
        }
        ncn += all;
    }
    if (threadIdx.x == 0) {
        *cc = ncn;
        *status_word = st;
    }
}

// LDS (dynamic): [map h*w T][cubic 16 f32][LimbLdsPy]
template <typename T>
__global__ __launch_bounds__(kThreads) void k_limb_connect_py(const T *__restrict__ net, int n_samples, int h, int w,
                                                              int flip, int maxp, int cap, int img_height,
                                                              const int *__restrict__ img_height_dev,
                                                              const float4 *__restrict__ peaks,
                                                              const int *__restrict__ counts, double4 *__restrict__ conns,
                                                              int *__restrict__ conn_counts,
                                                              unsigned *__restrict__ status) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int limb = blockIdx.x, img = blockIdx.y;
    const int pa = d_limb_pairs[limb][0], pb = d_limb_pairs[limb][1];
    int nA = counts[img * PP_NUM_PART + pa], nB = counts[img * PP_NUM_PART + pb];
    nA = nA < maxp ? nA : maxp;
    nB = nB < maxp ? nB : maxp;
    int *cc = conn_counts + img * PP_NUM_LIMB + limb;
    if (nA == 0 || nB == 0) {
        if (threadIdx.x == 0) {
            *cc = 0;
            status[img * kFlagWords + PP_NUM_PART + limb] = 0u;
        }
        return;
    }
    const int npix = h * w;
    size_t off = 0;
    T *smap = reinterpret_cast<T *>(lds_raw);
    off += (sizeof(T) * (size_t)npix + 15) & ~(size_t)15;
    float *s_cub = reinterpret_cast<float *>(lds_raw + off);
    off += 64;
    LimbLdsPy L = carve_limb_lds_py(lds_raw + off, maxp, cap);
    if (threadIdx.x < 16) s_cub[threadIdx.x] = d_cubic4[threadIdx.x >> 2][threadIdx.x & 3];
    const float4 *pka = peaks + ((size_t)img * PP_NUM_PART + pa) * maxp;
    const float4 *pkb = peaks + ((size_t)img * PP_NUM_PART + pb) * maxp;
    for (int i = threadIdx.x; i < nA; i += kThreads) {
        const float4 p = pka[i];
        L.ax[i] = p.x;
        L.ay[i] = p.y;
        L.as[i] = p.z;
    }
    for (int i = threadIdx.x; i < nB; i += kThreads) {
        const float4 p = pkb[i];
        L.bx[i] = p.x;
        L.by[i] = p.y;
        L.bs[i] = p.z;
    }
    for (int i = threadIdx.x; i < maxp; i += kThreads) {
        L.usedA[i] = 0;
        L.usedB[i] = 0;
    }
    const size_t plane = (size_t)npix;
    const T *o0 = net + ((size_t)img * n_samples * PP_NUM_CH + limb) * plane;
    const T *o1 = net + (((size_t)img * n_samples + 1) * PP_NUM_CH + d_flip_paf_ord[limb]) * plane;
    load_channel(smap, o0, o1, h, w, flip != 0);
    __syncthreads();
    LdsBicubicSampler<T> smp{smap, s_cub, h, w, w};
    const int ih = img_height_dev ? img_height_dev[img] : img_height;

    connect_limb_py(smp, L, nA, nB, cap, maxp, ih, conns + ((size_t)img * PP_NUM_LIMB + limb) * maxp, cc,
                    status + img * kFlagWords + PP_NUM_PART + limb);
}

// Host-array form (utils.parse_skeletons.find_connections): the caller's up-sampled (H, W, C) map in global memory
__global__ __launch_bounds__(kThreads) void k_limb_connect_py_hwc(const float *__restrict__ paf, int H, int W, int C, int maxp,
                                                                  int cap, int img_height, const float4 *__restrict__ peaks,
                                                                  const int *__restrict__ counts, double4 *__restrict__ conns,
                                                                  int *__restrict__ conn_counts, unsigned *__restrict__ status) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int limb = blockIdx.x;
    const int pa = d_limb_pairs[limb][0], pb = d_limb_pairs[limb][1];
    int nA = counts[pa], nB = counts[pb];
    nA = nA < maxp ? nA : maxp;
    nB = nB < maxp ? nB : maxp;
    int *cc = conn_counts + limb;
    if (nA == 0 || nB == 0 || limb >= C) {
        if (threadIdx.x == 0) {
            *cc = 0;
            status[PP_NUM_PART + limb] = 0u;
        }
        return;
    }
    LimbLdsPy L = carve_limb_lds_py(lds_raw, maxp, cap);
    const float4 *pka = peaks + (size_t)pa * maxp;
    const float4 *pkb = peaks + (size_t)pb * maxp;
    for (int i = threadIdx.x; i < nA; i += kThreads) {
        const float4 p = pka[i];
        L.ax[i] = p.x;
        L.ay[i] = p.y;
        L.as[i] = p.z;
    }
    for (int i = threadIdx.x; i < nB; i += kThreads) {
        const float4 p = pkb[i];
        L.bx[i] = p.x;
        L.by[i] = p.y;
        L.bs[i] = p.z;
    }
    for (int i = threadIdx.x; i < maxp; i += kThreads) {
        L.usedA[i] = 0;
        L.usedB[i] = 0;
    }
    __syncthreads();
    GlobalHwcSampler smp{paf, H, W, C, limb};
    connect_limb_py(smp, L, nA, nB, cap, maxp, img_height, conns + (size_t)limb * maxp, cc, status + PP_NUM_PART + limb);
}

// find_humans, one wave per image, float64 person table in LDS: [s][k] = {id, score}; k = 18: {total, -1}; 19: {count, len}
constexpr int kMaxSkelPy = 128;
__host__ __device__ inline size_t assemble_py_lds_bytes(int maxp) {
    return (size_t)kMaxSkelPy * kSkelStride * 16 + (size_t)PP_NUM_PART * maxp * 16 + (size_t)PP_NUM_LIMB * maxp * 32;
}

// PK = float4: refactored path (integer-valued coordinates, int x / y in the record); PK = double4: original path
// (fractional coordinates: the record's x / y fields then hold FLOAT bit patterns and PP_ST_FLOAT_COORDS is set).
template <typename PK>
__global__ __launch_bounds__(64) void k_assemble_py(int maxp, int explicit_ids, const PK *__restrict__ peaks,
                                                    const int *__restrict__ counts, const double4 *__restrict__ conns,
                                                    const int *__restrict__ conn_counts, const unsigned *__restrict__ status,
                                                    int flag_first, pp_record *__restrict__ records,
                                                    double *__restrict__ persons_out,
                                                    int *__restrict__ n_persons_out) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int ntab = PP_NUM_PART * maxp;
    double *pid = reinterpret_cast<double *>(lds_raw);                 // [kMaxSkelPy][21]
    double *psc = pid + kMaxSkelPy * kSkelStride;
    double4 *s_conn = reinterpret_cast<double4 *>(psc + kMaxSkelPy * kSkelStride);  // {src_id, dst_id, score, len}
    float *line_x = reinterpret_cast<float *>(s_conn + PP_NUM_LIMB * maxp);
    float *line_y = line_x + ntab;
    float *line_s = line_y + ntab;
    constexpr bool kFloatCoords = sizeof(PK) == sizeof(double4);
    __shared__ int s_off[PP_NUM_PART + 1];
    __shared__ int s_cnt[PP_NUM_PART];
    __shared__ int s_coff[PP_NUM_LIMB + 1];
    __shared__ int s_merge;

    const int *cnt_g = counts + img * PP_NUM_PART;
    const PK *pk_g = peaks + (size_t)img * PP_NUM_PART * maxp;
    if (lane == 0) {
        int run = 0;
        for (int k = 0; k < PP_NUM_PART; k++) {
            int c = cnt_g[k];
            c = c < maxp ? c : maxp;
            s_cnt[k] = c;
            s_off[k] = run;
            run += c;
        }
        s_off[PP_NUM_PART] = run;
        run = 0;
        for (int l = 0; l < PP_NUM_LIMB; l++) {
            s_coff[l] = run;
            int c = conn_counts[img * PP_NUM_LIMB + l];
            run += c < maxp ? c : maxp;
        }
        s_coff[PP_NUM_LIMB] = run;
    }
    __syncthreads();
    const int n_peaks = s_off[PP_NUM_PART];
    for (int part = 0; part < PP_NUM_PART; part++) {  // joint_candidates: rows flattened in part order (:423)
        const int c = s_cnt[part], o = s_off[part];
        for (int r = lane; r < c; r += 64) {
            const PK p = pk_g[(size_t)part * maxp + r];
            line_x[o + r] = (float)p.x;
            line_y[o + r] = (float)p.y;
            line_s[o + r] = (float)p.z;
        }
    }
    for (int limb = 0; limb < PP_NUM_LIMB; limb++) {
        const int c = s_coff[limb + 1] - s_coff[limb], o = s_coff[limb];
        const int part1 = d_limb_pairs[limb][0], part2 = d_limb_pairs[limb][1];
        const double4 *cn_g = conns + ((size_t)img * PP_NUM_LIMB + limb) * maxp;
        for (int ci = lane; ci < c; ci += 64) {
            double4 cn = cn_g[ci];
            if (!explicit_ids) {  // peak id == position in the part-ordered joint list
                cn.x = (double)(s_off[part1] + (int)cn.x);
                cn.y = (double)(s_off[part2] + (int)cn.y);
            }
            s_conn[o + ci] = cn;
        }
    }
    __syncthreads();

    int np = 0;  // uniform
    unsigned st = 0;
    for (int limb = 0; limb < PP_NUM_LIMB; limb++) {
        const int part1 = d_limb_pairs[limb][0], part2 = d_limb_pairs[limb][1];
        for (int ci = s_coff[limb]; ci < s_coff[limb + 1]; ci++) {
            const double4 cn = s_conn[ci];
            const double src_id = cn.x, dst_id = cn.y, score = cn.z, limb_len = cn.w;
            int num_found = 0, idx1 = 0, idx2 = 0;
            for (int base = 0; base < np; base += 64) {  // :440-450; matches beyond the second are ignored
                const int s = base + lane;
                bool hit = false;
                if (s < np) hit = (pid[s * kSkelStride + part1] == src_id) || (pid[s * kSkelStride + part2] == dst_id);
                unsigned long long m = __ballot(hit);
                if (m) {
                    if (num_found == 0) {
                        idx1 = base + __ffsll((long long)m) - 1;
                        const unsigned long long m2 = m & (m - 1);
                        if (m2) idx2 = base + __ffsll((long long)m2) - 1;
                    } else if (num_found == 1) {
                        idx2 = base + __ffsll((long long)m) - 1;
                    }
                    num_found += __popcll(m);
                }
            }
            if (num_found > 2) num_found = 2;
            const int isrc = (int)src_id, idst = (int)dst_id;  // joint_candidates[int(id), 2]: indexed BY ID (:474, :589)
            const double ps_src = (isrc >= 0 && isrc < n_peaks) ? (double)line_s[isrc] : 0.0;
            const double ps_dst = (idst >= 0 && idst < n_peaks) ? (double)line_s[idst] : 0.0;
            if (num_found == 1) {  // :452-487
                if (lane == 0) {
                    double *i1 = pid + idx1 * kSkelStride, *f1 = psc + idx1 * kSkelStride;
                    const double dpk = i1[part2], dsc = f1[part2], plen = f1[19];
                    const bool len_ok = __dmul_rn(plen, 16.0) > limb_len;
                    if ((int)dpk == -1 && len_ok) {
                        i1[part2] = dst_id;
                        f1[part2] = score;
                        i1[19] += 1.0;
                        f1[19] = limb_len > plen ? limb_len : plen;
                        i1[18] = __dadd_rn(i1[18], __dadd_rn(ps_dst, score));
                    } else if (((int)dpk != (int)dst_id && dsc <= score && len_ok) || ((int)dpk == (int)dst_id && dsc <= score)) {
                        // the OLD peak's score and the OLD limb score are subtracted first (:477-480)
                        const int old = (int)dpk;
                        const double old_ps = (old >= 0 && old < n_peaks) ? (double)line_s[old] : 0.0;
                        i1[18] = __dadd_rn(i1[18], -__dadd_rn(old_ps, dsc));
                        i1[part2] = dst_id;
                        f1[part2] = score;
                        f1[19] = limb_len > plen ? limb_len : plen;
                        i1[18] = __dadd_rn(i1[18], __dadd_rn(ps_dst, score));
                    }
                }
                __syncthreads();
            } else if (num_found == 2) {  // :489-560
                if (lane == 0) {
                    double *i1 = pid + idx1 * kSkelStride, *f1 = psc + idx1 * kSkelStride;
                    double *i2 = pid + idx2 * kSkelStride, *f2 = psc + idx2 * kSkelStride;
                    const double plen = f1[19];
                    bool shared = false, have1 = false, have2 = false;
                    double min1 = 0, min2 = 0;
                    for (int k = 0; k < PP_NUM_PART; k++) {
                        const bool m1 = i1[k] >= 0, m2 = i2[k] >= 0;  // :502-503
                        if (m1 && m2) shared = true;
                        if (m1 && (!have1 || f1[k] < min1)) { min1 = f1[k]; have1 = true; }
                        if (m2 && (!have2 || f2[k] < min2)) { min2 = f2[k]; have2 = true; }
                    }
                    int merge = 0;
                    if (!shared) {
                        const double mt = min1 < min2 ? min1 : min2;
                        if (score >= __dmul_rn(0.7, mt) && limb_len < __dmul_rn(plen, 16.0)) {  // :511-512 AND
                            for (int k = 0; k < PP_NUM_PART; k++) {  // np.maximum on (18, 2), :516
                                if (i2[k] > i1[k]) i1[k] = i2[k];
                                if (f2[k] > f1[k]) f1[k] = f2[k];
                            }
                            i1[19] += i2[19];
                            f1[19] = limb_len > plen ? limb_len : plen;
                            i1[18] = __dadd_rn(i1[18], __dadd_rn(i2[18], score));
                            merge = 1;
                        }
                    }
                    s_merge = merge;
                }
                __syncthreads();
                if (s_merge) {  // np.delete(person2)
                    for (int s = idx2; s < np - 1; s++) {
                        if (lane < 20) {
                            pid[s * kSkelStride + lane] = pid[(s + 1) * kSkelStride + lane];
                            psc[s * kSkelStride + lane] = psc[(s + 1) * kSkelStride + lane];
                        }
                    }
                    np--;
                }
                __syncthreads();
            } else {  // new person, :583-596
                if (np < kMaxSkelPy) {
                    if (lane < 20) {
                        double idv = -1.0, scv = -1.0;
                        if (lane == part1) { idv = src_id; scv = score; }
                        if (lane == part2) { idv = dst_id; scv = score; }
                        if (lane == 19) { idv = 2.0; scv = limb_len; }
                        if (lane == 18) idv = __dadd_rn(__dadd_rn(ps_src, ps_dst), score);
                        pid[np * kSkelStride + lane] = idv;
                        psc[np * kSkelStride + lane] = scv;
                    }
                    np++;
                } else {
                    st |= PP_ST_SKEL_OVERFLOW;
                }
                __syncthreads();
            }
        }
    }
    // ---- prune (:599-603) and records (evaluate.py:132-156: x, y, score from joint_candidates; score = total / count)
    pp_record *rec = records + img;
    int n_out = 0;
    for (int base = 0; base < np; base += 64) {
        const int s = base + lane;
        bool keep = false;
        if (s < np) {
            const double count = pid[s * kSkelStride + 19], total = pid[s * kSkelStride + 18];
            keep = !(count < 2.0 || total / count < 0.45);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int r = n_out + __popcll(m & lanemask_lt());
            if (r < PP_MAX_HUMANS) {
                pp_human *hm = rec->humans + r;
                for (int kp = 0; kp < PP_NUM_PART; kp++) {
                    const int id = (int)pid[s * kSkelStride + kp];
                    hm->peak_id[kp] = id;
                    const bool ok = id >= 0 && id < n_peaks;
                    const float fx = ok ? line_x[id] : 0.0f, fy = ok ? line_y[id] : 0.0f;
                    hm->x[kp] = kFloatCoords ? __float_as_int(fx) : (int)fx;
                    hm->y[kp] = kFloatCoords ? __float_as_int(fy) : (int)fy;
                    hm->part_score[kp] = ok ? line_s[id] : 0.0f;
                }
                hm->score = (float)(pid[s * kSkelStride + 18] / pid[s * kSkelStride + 19]);
                hm->n_parts = (int)pid[s * kSkelStride + 19];
            }
            if (persons_out) {  // raw person_to_joint_assoc rows (20, 2) float64
                double *row = persons_out + (size_t)r * 40;
                for (int k = 0; k < 20; k++) {
                    row[2 * k] = pid[s * kSkelStride + k];
                    row[2 * k + 1] = psc[s * kSkelStride + k];
                }
            }
        }
        n_out += __popcll(m);
    }
    if (lane == 0 && n_persons_out) *n_persons_out = n_out;
    const unsigned fl = or_flags(status, img, flag_first, lane);
    if (lane == 0) {
        if (n_out > PP_MAX_HUMANS) {
            st |= PP_ST_HUMAN_OVERFLOW;
            n_out = PP_MAX_HUMANS;
        }
        rec->n_humans = n_out;
        rec->n_peaks = n_peaks;
        rec->n_connections = s_coff[PP_NUM_LIMB];
        rec->status = fl | st | (kFloatCoords ? PP_ST_FLOAT_COORDS : 0u);
    }
}

// ================================================================================================ A10: original path
// predict (utils/parse_skeletons.py:180-283): maps of every scale are up-sampled x4, cropped, resized to the image size
// and averaged in float64; find_peaks (:286-321): 3x3 / >= thre1 NMS + refine_centroid at IMAGE resolution; then the
// Python twins on the float64 limb maps.  Maps of that size (512 x 512 x 50 x 8 B = 105 MB per image) live in HBM.

// OpenCV interpolateCubic (A = -0.75) in float with separately rounded operations (oracle: orc_cubic_coeffs)
__device__ __forceinline__ void cubic_coeffs(float x, float c[4]) {
    const float A = -0.75f;
    const float xp = __fadd_rn(x, 1.0f);
    c[0] = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(A, xp), 3.75f), xp), -6.0f), xp), 3.0f);
    c[1] = __fadd_rn(__fmul_rn(__fmul_rn(__fadd_rn(__fmul_rn(1.25f, x), -2.25f), x), x), 1.0f);
    const float y = __fadd_rn(1.0f, -x);
    c[2] = __fadd_rn(__fmul_rn(__fmul_rn(__fadd_rn(__fmul_rn(1.25f, y), -2.25f), y), y), 1.0f);
    c[3] = __fadd_rn(__fadd_rn(__fadd_rn(1.0f, -c[0]), -c[1]), -c[2]);
}

// Generic cv2.resize(INTER_CUBIC) of planar float maps: src (C, src_h_alloc, src_ld) of which rows [0, ch) x cols [0, cw)
// are the image (crop), dst (C, dh, dw).  ACC: dst is float64 and receives dst += (double)(value / n) (:280-281).
template <bool ACC>
__global__ __launch_bounds__(256) void k_resize_cubic(const float *__restrict__ src, long src_plane, int src_ld, int ch, int cw,
                                                      void *__restrict__ dst_, int C, int dh, int dw, double scale_x,
                                                      double scale_y, float n_div, int identity) {
    const long total = (long)C * dh * dw;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int dx = (int)(i % dw);
        long t = i / dw;
        const int dy = (int)(t % dh);
        const int c = (int)(t / dh);
        const float *S = src + (long)c * src_plane;
        float v;
        if (identity) {  // cv2.resize copies when the size does not change
            v = S[(long)dy * src_ld + dx];
        } else {
            float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
            const int sx = (int)floorf(fx);
            fx = __fadd_rn(fx, -(float)sx);
            float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
            const int sy = (int)floorf(fy);
            fy = __fadd_rn(fy, -(float)sy);
            float a[4], b[4];
            cubic_coeffs(fx, a);
            cubic_coeffs(fy, b);
            const int x0 = clampi(sx - 1, 0, cw - 1), x1 = clampi(sx, 0, cw - 1), x2 = clampi(sx + 1, 0, cw - 1),
                      x3 = clampi(sx + 2, 0, cw - 1);
            float hrow[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float *row = S + (long)clampi(sy - 1 + k, 0, ch - 1) * src_ld;
                float h = __fmul_rn(row[x0], a[0]);
                h = __fadd_rn(h, __fmul_rn(row[x1], a[1]));
                h = __fadd_rn(h, __fmul_rn(row[x2], a[2]));
                h = __fadd_rn(h, __fmul_rn(row[x3], a[3]));
                hrow[k] = h;
            }
            v = __fmul_rn(hrow[0], b[0]);
            v = __fadd_rn(v, __fmul_rn(hrow[1], b[1]));
            v = __fadd_rn(v, __fmul_rn(hrow[2], b[2]));
            v = __fadd_rn(v, __fmul_rn(hrow[3], b[3]));
        }
        if (ACC) {
            double *D = static_cast<double *>(dst_);
            D[i] = __dadd_rn(D[i], (double)(v / n_div));
        } else {
            static_cast<float *>(dst_)[i] = v;
        }
    }
}

// flip-average to PLANAR float32 (B, 50, h, w) (the HWC form of posepaf_epilogue.hip is for host callers)
template <typename T>
__global__ __launch_bounds__(256) void k_flip_average_planar(const T *__restrict__ net, int batch, int h, int w, int flip,
                                                             float *__restrict__ out) {
    const long plane = (long)h * w;
    const long total = (long)batch * PP_NUM_CH * plane;
    const long stride = (long)gridDim.x * blockDim.x;
    const int ns = flip ? 2 : 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % w);
        long t = i / w;
        const int y = (int)(t % h);
        t /= h;
        const int c = (int)(t % PP_NUM_CH);
        const long b = t / PP_NUM_CH;
        const int cf = c < PP_NUM_LIMB ? d_flip_paf_ord[c] : PP_NUM_LIMB + d_flip_heat_ord[c - PP_NUM_LIMB];
        const long i0 = ((b * ns) * PP_NUM_CH + c) * plane + (long)y * w + x;
        const long i1 = ((b * ns + 1) * PP_NUM_CH + cf) * plane + (long)y * w + (w - 1 - x);
        float v;
        if (sizeof(T) == 2) {
            const __half *p = reinterpret_cast<const __half *>(net);
            v = flip ? __half2float(__hmul(__hadd(p[i0], p[i1]), __float2half(0.5f))) : __half2float(p[i0]);
        } else {
            const float *p = reinterpret_cast<const float *>(net);
            v = flip ? __fadd_rn(p[i0], p[i1]) / 2.0f : p[i0];
        }
        out[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ A10: all scales, one pass
// predict's loop body for EVERY scale in one launch (utils/parse_skeletons.py:250-281): per output pixel of the image-sized
// accumulator, for scale 1, 2, ...: flip-average -> x4 bicubic -> crop -> bicubic resize to the image -> acc += value / n
// (float64, the reference's order), the accumulator living in a register until its ONE store.  The chain of round 2 wrote and
// re-read a float32 copy of every up-sampled map (118 MB per image at scale 1.5) and read-modify-wrote the 105 MB of
// accumulators once per scale: ~1 GB of HBM traffic per image against 23 MB of network output in and 105 MB out here.
// Workgroup = a 32 x 32 tile of ONE channel of one image.  Per scale the tile's pre-images are staged in LDS: the rows/columns
// of the cropped x4 map U its 4 x 4 taps reach, and the rows/columns of the flip-averaged network output A those reach.  Every
// value is computed by the same expressions (cubic_coeffs, products and sums rounded separately, left to right) as
// k_flip_average_planar / k_resize_cubic, so the accumulators are bit-identical to the chain's.
constexpr int kAccTile = 32, kAccMaxScales = 6;
struct AccScale {
    const void *net;       // (B, 2|1, 50, h, w)
    int h, w, ch, cw;      // map size; cropped size of its x4 up-sampling (4h - pad_down, 4w - pad_right)
    int identity;          // the cropped map already has the image size: cv2.resize copies
    double sx, sy;         // cw / img_w, ch / img_h
};
struct AccParams {
    AccScale s[kAccMaxScales];
    int n, flip, img_h, img_w, u_cap, a_cap;   // u_cap / a_cap: LDS tile capacities in elements
    float n_div;
    double *heat_acc, *paf_acc;
};
__device__ __forceinline__ int src_floor(int d, double scale) { return (int)floorf((float)(((double)d + 0.5) * scale - 0.5)); }

template <typename T>
__global__ __launch_bounds__(256) void k_accumulate_scales(const AccParams P) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float *sU = reinterpret_cast<float *>(lds_raw);
    float *sA = sU + P.u_cap;
    const int tiles_x = (P.img_w + kAccTile - 1) / kAccTile;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int c = blockIdx.y, b = blockIdx.z;
    const int oy0 = ty * kAccTile, ox0 = tx * kAccTile;
    const int oy1 = min(oy0 + kAccTile, P.img_h), ox1 = min(ox0 + kAccTile, P.img_w);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // this thread's output pixels: (oy0 + ly + 8 k, ox0 + lx)
    const int ns = P.flip ? 2 : 1;
    const int cf = c < PP_NUM_LIMB ? d_flip_paf_ord[c] : PP_NUM_LIMB + d_flip_heat_ord[c - PP_NUM_LIMB];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int si = 0; si < P.n; si++) {
        const AccScale &S = P.s[si];
        // ---- rows / columns of U the tile's taps reach, then rows / columns of A those reach (clamped like the taps are)
        int ur0, ur1, uc0, uc1;
        if (S.identity) {
            ur0 = oy0, ur1 = oy1 - 1, uc0 = ox0, uc1 = ox1 - 1;
        } else {
            ur0 = clampi(src_floor(oy0, S.sy) - 1, 0, S.ch - 1), ur1 = clampi(src_floor(oy1 - 1, S.sy) + 2, 0, S.ch - 1);
            uc0 = clampi(src_floor(ox0, S.sx) - 1, 0, S.cw - 1), uc1 = clampi(src_floor(ox1 - 1, S.sx) + 2, 0, S.cw - 1);
        }
        const int un = ur1 - ur0 + 1, um = uc1 - uc0 + 1;
        const int ar0 = clampi(src_floor(ur0, 0.25) - 1, 0, S.h - 1), ar1 = clampi(src_floor(ur1, 0.25) + 2, 0, S.h - 1);
        const int ac0 = clampi(src_floor(uc0, 0.25) - 1, 0, S.w - 1), ac1 = clampi(src_floor(uc1, 0.25) + 2, 0, S.w - 1);
        const int an = ar1 - ar0 + 1, am = ac1 - ac0 + 1;
        // ---- A tile: flip-average (utils/parse_skeletons.py:230-236) in the array's dtype, widened
        const long plane = (long)S.h * S.w;
        const T *o0 = static_cast<const T *>(S.net) + (((long)b * ns) * PP_NUM_CH + c) * plane;
        const T *o1 = static_cast<const T *>(S.net) + (((long)b * ns + 1) * PP_NUM_CH + cf) * plane;
        for (int i = threadIdx.x; i < an * am; i += 256) {
            const int y = ar0 + i / am, x = ac0 + i % am;
            float v;
            if (sizeof(T) == 2) {
                const __half *p0 = reinterpret_cast<const __half *>(o0), *p1 = reinterpret_cast<const __half *>(o1);
                v = P.flip ? __half2float(__hmul(__hadd(p0[y * S.w + x], p1[y * S.w + (S.w - 1 - x)]), __float2half(0.5f)))
                           : __half2float(p0[y * S.w + x]);
            } else {
                const float *p0 = reinterpret_cast<const float *>(o0), *p1 = reinterpret_cast<const float *>(o1);
                v = P.flip ? __fadd_rn(p0[y * S.w + x], p1[y * S.w + (S.w - 1 - x)]) / 2.0f : p0[y * S.w + x];
            }
            sA[i] = v;
        }
        __syncthreads();
        // ---- U tile: cv2.resize(fx = fy = 4, INTER_CUBIC) of A (:252-263), taps clamped to the map
        for (int i = threadIdx.x; i < un * um; i += 256) {
            const int uy = ur0 + i / um, ux = uc0 + i % um;
            float fx = (float)(((double)ux + 0.5) * 0.25 - 0.5);
            const int sx = (int)floorf(fx);
            fx = __fadd_rn(fx, -(float)sx);
            float fy = (float)(((double)uy + 0.5) * 0.25 - 0.5);
            const int sy = (int)floorf(fy);
            fy = __fadd_rn(fy, -(float)sy);
            float ca[4], cb[4];
            cubic_coeffs(fx, ca);
            cubic_coeffs(fy, cb);
            const int x0 = clampi(sx - 1, 0, S.w - 1) - ac0, x1 = clampi(sx, 0, S.w - 1) - ac0, x2 = clampi(sx + 1, 0, S.w - 1) - ac0,
                      x3 = clampi(sx + 2, 0, S.w - 1) - ac0;
            float hrow[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float *row = sA + (clampi(sy - 1 + k, 0, S.h - 1) - ar0) * am;
                float hv = __fmul_rn(row[x0], ca[0]);
                hv = __fadd_rn(hv, __fmul_rn(row[x1], ca[1]));
                hv = __fadd_rn(hv, __fmul_rn(row[x2], ca[2]));
                hv = __fadd_rn(hv, __fmul_rn(row[x3], ca[3]));
                hrow[k] = hv;
            }
            float v = __fmul_rn(hrow[0], cb[0]);
            v = __fadd_rn(v, __fmul_rn(hrow[1], cb[1]));
            v = __fadd_rn(v, __fmul_rn(hrow[2], cb[2]));
            v = __fadd_rn(v, __fmul_rn(hrow[3], cb[3]));
            sU[i] = v;
        }
        __syncthreads();
        // ---- crop (:272-273), resize to the image size (:276-277), += value / n in float64 (:280-281)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int dy = oy0 + ly + 8 * k, dx = ox0 + lx;
            if (dy < oy1 && dx < ox1) {
                float v;
                if (S.identity) {
                    v = sU[(dy - ur0) * um + (dx - uc0)];
                } else {
                    float fx = (float)(((double)dx + 0.5) * S.sx - 0.5);
                    const int sx = (int)floorf(fx);
                    fx = __fadd_rn(fx, -(float)sx);
                    float fy = (float)(((double)dy + 0.5) * S.sy - 0.5);
                    const int sy = (int)floorf(fy);
                    fy = __fadd_rn(fy, -(float)sy);
                    float ca[4], cb[4];
                    cubic_coeffs(fx, ca);
                    cubic_coeffs(fy, cb);
                    const int x0 = clampi(sx - 1, 0, S.cw - 1) - uc0, x1 = clampi(sx, 0, S.cw - 1) - uc0,
                              x2 = clampi(sx + 1, 0, S.cw - 1) - uc0, x3 = clampi(sx + 2, 0, S.cw - 1) - uc0;
                    float hrow[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float *row = sU + (clampi(sy - 1 + r, 0, S.ch - 1) - ur0) * um;
                        float hv = __fmul_rn(row[x0], ca[0]);
                        hv = __fadd_rn(hv, __fmul_rn(row[x1], ca[1]));
                        hv = __fadd_rn(hv, __fmul_rn(row[x2], ca[2]));
                        hv = __fadd_rn(hv, __fmul_rn(row[x3], ca[3]));
                        hrow[r] = hv;
                    }
                    v = __fmul_rn(hrow[0], cb[0]);
                    v = __fadd_rn(v, __fmul_rn(hrow[1], cb[1]));
                    v = __fadd_rn(v, __fmul_rn(hrow[2], cb[2]));
                    v = __fadd_rn(v, __fmul_rn(hrow[3], cb[3]));
                }
                acc[k] = __dadd_rn(acc[k], (double)(v / P.n_div));
            }
        }
        __syncthreads();   // the tiles are rewritten by the next scale
    }
    double *D = c < PP_NUM_LIMB ? P.paf_acc + ((long)b * PP_NUM_LIMB + c) * P.img_h * P.img_w
                                : P.heat_acc + ((long)b * PP_NUM_HEAT + (c - PP_NUM_LIMB)) * P.img_h * P.img_w;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dy = oy0 + ly + 8 * k, dx = ox0 + lx;
        if (dy < oy1 && dx < ox1) D[(long)dy * P.img_w + dx] = acc[k];
    }
}

// find_peaks at image resolution: one workgroup per (part, image).  The float64 accumulator is cast to float32 on read
// (:290); 3x3 / >= thre NMS (reflect padding == ignore out-of-map neighbours); np.nonzero order via per-thread contiguous
// pixel ranges + block scan; refine_centroid per peak.  peaks: double4 (x, y, score, id-unused).
__global__ __launch_bounds__(kThreads) void k_fullres_peaks(const double *__restrict__ heat_acc, int H, int W, float thre,
                                                            int maxp, unsigned char *__restrict__ mask_scratch,
                                                            double4 *__restrict__ peaks, int *__restrict__ counts,
                                                            unsigned *__restrict__ status) {
    __shared__ int s_wsum[kWaves];
    __shared__ int s_pk[PP_MAX_PEAKS_PER_PART_LIMIT];
    const int part = blockIdx.x, img = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long npix = (long)H * W;
    const double *M = heat_acc + ((long)img * PP_NUM_HEAT + part) * npix;
    unsigned char *mask = mask_scratch + ((long)img * PP_NUM_PART + part) * npix;
    auto val = [&](long i) -> float { return (float)M[i]; };
    for (long i = threadIdx.x; i < npix; i += kThreads) {
        const float v = val(i);
        unsigned char pk = 0;
        if (v >= thre) {
            const int y = (int)(i / W), x = (int)(i - (long)y * W);
            pk = 1;
            for (int dy = -1; dy <= 1 && pk; dy++) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    if (val((long)yy * W + xx) > v) {
                        pk = 0;
                        break;
                    }
                }
            }
        }
        mask[i] = pk;
    }
    __syncthreads();  // same workgroup: its global writes are visible to it after the barrier
    const long ppt = (npix + kThreads - 1) / kThreads;
    const long b0 = (long)threadIdx.x * ppt;
    int cnt = 0;
    for (long q = b0; q < b0 + ppt && q < npix; q++) cnt += mask[q];
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(incl, d);
        if (lane >= d) incl += n;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int rank = incl - cnt;
    for (int q = 0; q < wave; q++) rank += s_wsum[q];
    int total = 0;
    for (int q = 0; q < kWaves; q++) total += s_wsum[q];
    for (long q = b0; q < b0 + ppt && q < npix && rank < maxp; q++)
        if (mask[q]) s_pk[rank++] = (int)q;
    __syncthreads();
    const int kept = total < maxp ? total : maxp;
    double4 *out = peaks + ((long)img * PP_NUM_PART + part) * maxp;
    for (int p = wave; p < kept; p += kWaves) {  // refine_centroid, radius 2 (utils/util.py:188-213)
        const int i = s_pk[p];
        const int py = i / W, px = i - py * W;
        double ox, oy, sc;
        if (py - 2 < 0 || py + 3 > H || px - 2 < 0 || px + 3 > W) {
            ox = (double)px;
            oy = (double)py;
            sc = (double)val(i);
        } else {
            double sx = 0.0, sy = 0.0, sv = 0.0;
            if (lane < 25) {
                const int r = lane / 5, c = lane - r * 5;
                const double v = (double)val((long)(py - 2 + r) * W + (px - 2 + c));
                sx = v * (double)(r - 2);  // x_grid varies along rows (np.mgrid): restated as written
                sy = v * (double)(c - 2);
                sv = v;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                sx += __shfl_xor(sx, d);
                sy += __shfl_xor(sy, d);
                sv += __shfl_xor(sv, d);
            }
            ox = (double)px + sx / sv;
            oy = (double)py + sy / sv;
            sc = sv / 25.0;
        }
        if (lane == 0) out[p] = make_double4(ox, oy, (double)(float)sc, 0.0);  // box.mean() is a float32 scalar
    }
    if (threadIdx.x == 0) {
        counts[img * PP_NUM_PART + part] = total;
        status[img * kFlagWords + part] = total > maxp ? PP_ST_PEAK_OVERFLOW : 0u;  // plain store, every launch
    }
}

struct GlobalPlanarF64Sampler {  // predict's paf_avg, planar (30, H, W) float64
    const double *paf;
    int H, W;
    __device__ __forceinline__ double at(int X, int Y) const {
        X = clampi(X, 0, W - 1);
        Y = clampi(Y, 0, H - 1);
        return paf[(long)Y * W + X];
    }
};

__global__ __launch_bounds__(kThreads) void k_limb_connect_py_fullres(const double *__restrict__ paf_acc, int H, int W, int maxp,
                                                                      int cap, int img_height,
                                                                      const double4 *__restrict__ peaks,
                                                                      const int *__restrict__ counts, double4 *__restrict__ conns,
                                                                      int *__restrict__ conn_counts, unsigned *__restrict__ status) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int limb = blockIdx.x, img = blockIdx.y;
    const int pa = d_limb_pairs[limb][0], pb = d_limb_pairs[limb][1];
    int nA = counts[img * PP_NUM_PART + pa], nB = counts[img * PP_NUM_PART + pb];
    nA = nA < maxp ? nA : maxp;
    nB = nB < maxp ? nB : maxp;
    int *cc = conn_counts + img * PP_NUM_LIMB + limb;
    if (nA == 0 || nB == 0) {
        if (threadIdx.x == 0) {
            *cc = 0;
            status[img * kFlagWords + PP_NUM_PART + limb] = 0u;
        }
        return;
    }
    LimbLdsPy L = carve_limb_lds_py(lds_raw, maxp, cap);
    const double4 *pka = peaks + ((long)img * PP_NUM_PART + pa) * maxp;
    const double4 *pkb = peaks + ((long)img * PP_NUM_PART + pb) * maxp;
    for (int i = threadIdx.x; i < nA; i += kThreads) {
        const double4 p = pka[i];
        L.ax[i] = p.x;
        L.ay[i] = p.y;
        L.as[i] = p.z;
    }
    for (int i = threadIdx.x; i < nB; i += kThreads) {
        const double4 p = pkb[i];
        L.bx[i] = p.x;
        L.by[i] = p.y;
        L.bs[i] = p.z;
    }
    for (int i = threadIdx.x; i < maxp; i += kThreads) {
        L.usedA[i] = 0;
        L.usedB[i] = 0;
    }
    __syncthreads();
    GlobalPlanarF64Sampler smp{paf_acc + ((long)img * PP_NUM_LIMB + limb) * (long)H * W, H, W};
    connect_limb_py(smp, L, nA, nB, cap, maxp, img_height, conns + ((long)img * PP_NUM_LIMB + limb) * maxp, cc,
                    status + img * kFlagWords + PP_NUM_PART + limb);
}

// ------------------------------------------------------------------------------------------------ launchers
size_t lds_bytes_heat(int elem, int h, int w, int maxp) {
    const size_t npix = (size_t)h * w;
    return ((elem * npix + 15) & ~(size_t)15) + 64 + ((4 * (size_t)maxp + 15) & ~(size_t)15) + ((npix + 7) / 8 + 15) / 16 * 16;
}
size_t lds_bytes_limb(int elem, int h, int w, int maxp, int cap) {
    const size_t npix = (size_t)h * (w + 16 / elem);   // padded rows: limb_map_ld
    const size_t limb = ((elem * npix + 15) & ~(size_t)15) + 64 + limb_lds_bytes(maxp, cap);
    const size_t tail = assemble_wave_lds_bytes(maxp);  // the last limb workgroup of an image assembles it in the same region
    return limb > tail ? limb : tail;
}
size_t lds_bytes_limb_hwc(int maxp, int cap) { return limb_lds_bytes(maxp, cap); }
size_t lds_bytes_assemble(int maxp) { return assemble_lds_bytes(maxp); }

// Dynamic LDS above the 64 KB default needs the attribute; set once per process (pp_create), not per launch, so
// that the per-batch entry points stay free of anything but kernel launches (hipGraph-capturable).
hipError_t set_stamp_buffer(long long *buf) {
    const char *e = getenv("POSEPAF_STAMP_REALTIME");
    const int rt = e && e[0] == '1';
    hipError_t err = hipMemcpyToSymbol(HIP_SYMBOL(d_stamp_realtime), &rt, sizeof(rt));
    if (err != hipSuccess) return err;
    return hipMemcpyToSymbol(HIP_SYMBOL(d_stamps), &buf, sizeof(buf));
}

hipError_t init_kernel_attributes() {
    const int lim = (int)kMaxDynLds;
    if (const char *e = getenv("POSEPAF_KB_SP_PAIRS")) {   // A/B measurements of the sample-parallel threshold
        int v = atoi(e);
        v = v < 0 ? 0 : (v > 512 ? 512 : v);
        hipError_t err = hipMemcpyToSymbol(HIP_SYMBOL(d_sp_max_pairs), &v, sizeof(v));
        if (err != hipSuccess) return err;
    }
    const void *fns[] = {reinterpret_cast<const void *>(&k_heat_peaks<__half>),
                         reinterpret_cast<const void *>(&k_heat_peaks<float>),
                         reinterpret_cast<const void *>(&k_limb_connect<__half, 256>),
                         reinterpret_cast<const void *>(&k_limb_connect<float, 256>),
                         reinterpret_cast<const void *>(&k_limb_connect<__half, 512>),
                         reinterpret_cast<const void *>(&k_limb_connect<float, 512>),
                         reinterpret_cast<const void *>(&k_limb_connect_hwc),
                         reinterpret_cast<const void *>(&k_assemble),
                         reinterpret_cast<const void *>(&k_assemble_wave),
                         reinterpret_cast<const void *>(&k_limb_connect_py<__half>),
                         reinterpret_cast<const void *>(&k_limb_connect_py<float>),
                         reinterpret_cast<const void *>(&k_limb_connect_py_hwc),
                         reinterpret_cast<const void *>(&k_assemble_py<float4>),
                         reinterpret_cast<const void *>(&k_assemble_py<double4>),
                         reinterpret_cast<const void *>(&k_limb_connect_py_fullres)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_heat_peaks(const void *net, int dtype, int batch, int n_samples, int h, int w, int flip, int refine,
                             int nms_mode, float thr, int maxp, float4 *peaks, int *counts, unsigned *status, int *order,
                             int *arrive_all, hipStream_t stream) {
    const dim3 grid(PP_NUM_PART, batch), block(kThreads);
    const size_t lds = lds_bytes_heat(dtype == PP_F16 ? 2 : 4, h, w, maxp);
    if ((size_t)batch * sizeof(int) > lds) order = nullptr;  // the sorter keeps one int per image in the map's LDS region
    if (dtype == PP_F16)
        hipLaunchKernelGGL(k_heat_peaks<__half>, grid, block, lds, stream, static_cast<const __half *>(net), n_samples, h,
                           w, flip, refine, nms_mode, thr, maxp, peaks, counts, status, order, arrive_all);
    else
        hipLaunchKernelGGL(k_heat_peaks<float>, grid, block, lds, stream, static_cast<const float *>(net), n_samples, h, w,
                           flip, refine, nms_mode, thr, maxp, peaks, counts, status, order, arrive_all);
    return hipGetLastError();
}
bool heat_peaks_sorts(int dtype, int batch, int h, int w, int maxp) {
    return (size_t)batch * sizeof(int) <= lds_bytes_heat(dtype == PP_F16 ? 2 : 4, h, w, maxp);
}

// threads per limb workgroup: 256; POSEPAF_KB_THREADS=512 selects the 8-wave instance (A/B: 250 us against 182 us at 128 images
// per launch -- two workgroups per CU instead of three cost more than the shorter scoring of the crowded limbs gains)
int limb_threads() {
    static const int nt = [] {
        const char *e = getenv("POSEPAF_KB_THREADS");
        return (e && atoi(e) == 512) ? 512 : 256;
    }();
    return nt;
}

hipError_t launch_limb_connect(const void *net, int dtype, int batch, int n_samples, int h, int w, int flip, int maxp,
                               int cap, int min_img_size, const int *min_img_size_dev, const float4 *peaks,
                               const int *counts, float4 *conns, float4 *aux, int *conn_counts, unsigned *status,
                               const int *order, int *arrive, unsigned *ready, pp_record *records, hipStream_t stream) {
    const int nt = limb_threads();
    const dim3 grid(PP_NUM_LIMB + (arrive ? 1 : 0), batch), block(nt);   // fused form: workgroup 30 of an image assembles it
    const size_t lds = lds_bytes_limb(dtype == PP_F16 ? 2 : 4, h, w, maxp, cap);  // >= the assembly tail's need
#define PP_LAUNCH_LIMB(T, NT)                                                                                                   \
    hipLaunchKernelGGL((k_limb_connect<T, NT>), grid, block, lds, stream, static_cast<const T *>(net), n_samples, h, w, flip, maxp, \
                       cap, min_img_size, min_img_size_dev, peaks, counts, conns, aux, conn_counts, status, order, arrive, ready, records)
    if (dtype == PP_F16) {
        if (nt == 512) PP_LAUNCH_LIMB(__half, 512);
        else PP_LAUNCH_LIMB(__half, 256);
    } else {
        if (nt == 512) PP_LAUNCH_LIMB(float, 512);
        else PP_LAUNCH_LIMB(float, 256);
    }
#undef PP_LAUNCH_LIMB
    return hipGetLastError();
}

hipError_t launch_assemble_wave(int batch, int maxp, const float4 *peaks, const int *counts, const float4 *conns,
                                const float4 *aux, const int *conn_counts, const unsigned *status, pp_record *records,
                                hipStream_t stream) {
    hipLaunchKernelGGL(k_assemble_wave, dim3(batch), dim3(64), assemble_wave_lds_bytes(maxp), stream, maxp, peaks, counts, conns,
                       aux, conn_counts, status, records);
    return hipGetLastError();
}

hipError_t launch_limb_connect_hwc(const float *paf, int H, int W, int C, int maxp, int cap, int min_img_size,
                                   const float4 *peaks, const int *counts, float4 *conns, int *conn_counts,
                                   unsigned *status, hipStream_t stream) {
    const size_t lds = lds_bytes_limb_hwc(maxp, cap);
    hipLaunchKernelGGL(k_limb_connect_hwc, dim3(PP_NUM_LIMB), dim3(kThreads), lds, stream, paf, H, W, C, maxp, cap,
                       min_img_size, peaks, counts, conns, conn_counts, status);
    return hipGetLastError();
}

hipError_t launch_assemble(int batch, int maxp, int explicit_ids, const float4 *peaks, const int *counts,
                           const float4 *conns, const int *conn_counts, const unsigned *status, int flag_first,
                           pp_record *records, hipStream_t stream) {
    const size_t lds = lds_bytes_assemble(maxp);
    hipLaunchKernelGGL(k_assemble, dim3(batch), dim3(64), lds, stream, maxp, explicit_ids, peaks, counts, conns,
                       conn_counts, status, flag_first, records);
    return hipGetLastError();
}

size_t lds_bytes_limb_py(int elem, int h, int w, int maxp, int cap) {
    return ((elem * (size_t)h * w + 15) & ~(size_t)15) + 64 + limb_lds_bytes_py(maxp, cap);
}
size_t lds_bytes_assemble_py(int maxp) { return assemble_py_lds_bytes(maxp); }

hipError_t launch_limb_connect_py(const void *net, int dtype, int batch, int n_samples, int h, int w, int flip, int maxp,
                                  int cap, int img_height, const int *img_height_dev, const float4 *peaks, const int *counts,
                                  void *conns, int *conn_counts, unsigned *status, hipStream_t stream) {
    const dim3 grid(PP_NUM_LIMB, batch), block(kThreads);
    if (dtype == PP_F16) {
        hipLaunchKernelGGL(k_limb_connect_py<__half>, grid, block, lds_bytes_limb_py(2, h, w, maxp, cap), stream,
                           static_cast<const __half *>(net), n_samples, h, w, flip, maxp, cap, img_height, img_height_dev,
                           peaks, counts, static_cast<double4 *>(conns), conn_counts, status);
    } else {
        hipLaunchKernelGGL(k_limb_connect_py<float>, grid, block, lds_bytes_limb_py(4, h, w, maxp, cap), stream,
                           static_cast<const float *>(net), n_samples, h, w, flip, maxp, cap, img_height, img_height_dev,
                           peaks, counts, static_cast<double4 *>(conns), conn_counts, status);
    }
    return hipGetLastError();
}

hipError_t launch_assemble_py(int batch, int maxp, int explicit_ids, const float4 *peaks, const int *counts, const void *conns,
                              const int *conn_counts, const unsigned *status, int flag_first, pp_record *records,
                              double *persons_out, int *n_persons_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_assemble_py<float4>, dim3(batch), dim3(64), lds_bytes_assemble_py(maxp), stream, maxp, explicit_ids,
                       peaks, counts, static_cast<const double4 *>(conns), conn_counts, status, flag_first, records, persons_out,
                       n_persons_out);
    return hipGetLastError();
}

// ---- original path
hipError_t launch_resize_cubic(const float *src, long src_plane, int src_ld, int ch, int cw, void *dst, int acc, int C, int dh,
                               int dw, double scale_x, double scale_y, float n_div, hipStream_t stream) {
    const long total = (long)C * dh * dw;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const int identity = (ch == dh && cw == dw) ? 1 : 0;
    if (acc)
        hipLaunchKernelGGL(k_resize_cubic<true>, dim3(blocks), dim3(256), 0, stream, src, src_plane, src_ld, ch, cw, dst, C, dh, dw,
                           scale_x, scale_y, n_div, identity);
    else
        hipLaunchKernelGGL(k_resize_cubic<false>, dim3(blocks), dim3(256), 0, stream, src, src_plane, src_ld, ch, cw, dst, C, dh,
                           dw, scale_x, scale_y, n_div, identity);
    return hipGetLastError();
}

hipError_t launch_flip_average_planar(const void *net, int dtype, int batch, int h, int w, int flip, float *out,
                                      hipStream_t stream) {
    const long total = (long)batch * PP_NUM_CH * h * w;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (dtype == PP_F16)
        hipLaunchKernelGGL(k_flip_average_planar<__half>, dim3(blocks), dim3(256), 0, stream, static_cast<const __half *>(net),
                           batch, h, w, flip, out);
    else
        hipLaunchKernelGGL(k_flip_average_planar<float>, dim3(blocks), dim3(256), 0, stream, static_cast<const float *>(net),
                           batch, h, w, flip, out);
    return hipGetLastError();
}

// One launch for all scales; PP_ERR-like failure (hipErrorInvalidValue) when a scale's tiles do not fit LDS (the caller then
// uses the per-scale chain).  nets[i]: (batch, 2|1, 50, hs[i], ws[i]).
hipError_t launch_accumulate_scales(int n_scales, const void *const *nets, int dtype, int batch, const int *hs, const int *ws,
                                    int flip, const int *pad_down, const int *pad_right, int img_h, int img_w, double *heat_acc,
                                    double *paf_acc, hipStream_t stream) {
    if (n_scales < 1 || n_scales > kAccMaxScales) return hipErrorInvalidValue;
    AccParams P;
    P.n = n_scales;
    P.flip = flip;
    P.img_h = img_h;
    P.img_w = img_w;
    P.n_div = (float)n_scales;
    P.heat_acc = heat_acc;
    P.paf_acc = paf_acc;
    int umax = 0, amax = 0;
    for (int i = 0; i < n_scales; i++) {
        AccScale &S = P.s[i];
        S.net = nets[i];
        S.h = hs[i];
        S.w = ws[i];
        S.ch = 4 * hs[i] - pad_down[i];
        S.cw = 4 * ws[i] - pad_right[i];
        if (S.ch <= 0 || S.cw <= 0) return hipErrorInvalidValue;
        S.identity = (S.ch == img_h && S.cw == img_w) ? 1 : 0;
        S.sx = 1.0 / ((double)img_w / (double)S.cw);
        S.sy = 1.0 / ((double)img_h / (double)S.ch);
        const int ur = S.identity ? kAccTile : (int)(kAccTile * S.sy) + 6, uc = S.identity ? kAccTile : (int)(kAccTile * S.sx) + 6;
        const int ar = ur / 4 + 6, ac = uc / 4 + 6;
        umax = ur * uc > umax ? ur * uc : umax;
        amax = ar * ac > amax ? ar * ac : amax;
    }
    P.u_cap = umax;
    P.a_cap = amax;
    const size_t lds = ((size_t)umax + amax) * sizeof(float);
    if (lds > 60000) return hipErrorInvalidValue;
    const dim3 grid(((img_w + kAccTile - 1) / kAccTile) * ((img_h + kAccTile - 1) / kAccTile), PP_NUM_CH, batch);
    if (dtype == PP_F16) hipLaunchKernelGGL(k_accumulate_scales<__half>, grid, dim3(256), lds, stream, P);
    else hipLaunchKernelGGL(k_accumulate_scales<float>, grid, dim3(256), lds, stream, P);
    return hipGetLastError();
}

hipError_t launch_fullres(int batch, int H, int W, float thre1, int maxp, int cap, int img_height, const double *heat_acc,
                          const double *paf_acc, unsigned char *mask_scratch, void *peaks64, int *counts, void *conns,
                          int *conn_counts, unsigned *status, pp_record *records, hipStream_t stream) {
    hipLaunchKernelGGL(k_fullres_peaks, dim3(PP_NUM_PART, batch), dim3(kThreads), 0, stream, heat_acc, H, W, thre1, maxp,
                       mask_scratch, static_cast<double4 *>(peaks64), counts, status);
    hipLaunchKernelGGL(k_limb_connect_py_fullres, dim3(PP_NUM_LIMB, batch), dim3(kThreads), limb_lds_bytes_py(maxp, cap), stream,
                       paf_acc, H, W, maxp, cap, img_height, static_cast<const double4 *>(peaks64), counts,
                       static_cast<double4 *>(conns), conn_counts, status);
    hipLaunchKernelGGL(k_assemble_py<double4>, dim3(batch), dim3(64), lds_bytes_assemble_py(maxp), stream, maxp, 0,
                       static_cast<const double4 *>(peaks64), counts, static_cast<const double4 *>(conns), conn_counts, status,
                       0, records, static_cast<double *>(nullptr), static_cast<int *>(nullptr));
    return hipGetLastError();
}

hipError_t launch_limb_connect_py_hwc(const float *paf, int H, int W, int C, int maxp, int cap, int img_height,
                                      const float4 *peaks, const int *counts, void *conns, int *conn_counts, unsigned *status,
                                      hipStream_t stream) {
    hipLaunchKernelGGL(k_limb_connect_py_hwc, dim3(PP_NUM_LIMB), dim3(kThreads), limb_lds_bytes_py(maxp, cap), stream, paf, H,
                       W, C, maxp, cap, img_height, peaks, counts, static_cast<double4 *>(conns), conn_counts, status);
    return hipGetLastError();
}

}  // namespace pp
