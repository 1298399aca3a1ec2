/*
 * posepaf_oracle.c -- TEST INFRASTRUCTURE ONLY (see posepaf_oracle.h).
 *
 * Plain-C restatement of the reference's post-processing path.  Build with
 *   gcc -O2 -ffp-contract=off -fPIC -shared   (oracle/Makefile)
 * -ffp-contract=off matters: the reference is built by distutils with plain g++ on x86-64
 * (utils/pafprocess/setup.py:8-11), i.e. no fused multiply-add anywhere.
 *
 * All file:line citations are relative to /root/reference.
 */
#include "posepaf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables */

/* utils/pafprocess/pafprocess.h:21-27 == config/config.py:114-121 */
static const int LIMB_PAIRS[ORC_NUM_LIMB][2] = {
    {1, 0},   {1, 14},  {1, 15},  {1, 16},  {1, 17},  {0, 14},  {0, 15},  {14, 16}, {15, 17}, {1, 2},
    {2, 3},   {3, 4},   {1, 5},   {5, 6},   {6, 7},   {1, 8},   {8, 9},   {9, 10},  {1, 11},  {11, 12},
    {12, 13}, {0, 2},   {0, 5},   {2, 8},   {8, 12},  {5, 11},  {11, 9},  {16, 2},  {17, 5},  {8, 11}};

/* config/config.py:150-152 */
static const int FLIP_HEAT_ORD[ORC_NUM_HEAT] = {0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16, 18, 19};
static const int FLIP_PAF_ORD[ORC_NUM_LIMB] = {0,  2,  1,  4,  3,  6,  5,  8,  7,  12, 13, 14, 9,  10, 11,
                                               18, 19, 20, 15, 16, 17, 22, 21, 25, 26, 23, 24, 28, 27, 29};

const int *orc_limb_pairs(void) { return &LIMB_PAIRS[0][0]; }
const int *orc_flip_heat_ord(void) { return FLIP_HEAT_ORD; }
const int *orc_flip_paf_ord(void) { return FLIP_PAF_ORD; }

/* utils/pafprocess/pafprocess.h:6-18 */
static const float THRESH_PAF_SCORE = 0.1f;
static const float THRESH_PAF_STEP_RATIO = 0.8f;
static const int THRESH_PART_CNT = 2;
static const float THRESH_SKELETON_SCORE = 0.45f;
static const int STEP_PAF = 20;
static const int LIMB_LENGTH_RATE = 16;
static const float MIN_SCORE_TOLERANCE = 0.7f;
static const float PAF_OUT_WEIGHTS[3] = {0.5f, 0.25f, 0.25f};
#define NOT_ASSIGNED (-1)
#define SCORE_IDX ORC_NUM_PART          /* pafprocess.cpp:12 */
#define LIMB_INFO_IDX (ORC_NUM_PART + 1) /* pafprocess.cpp:11 */
#define NUM_PART_OUTS (ORC_NUM_PART + 2)

/* ------------------------------------------------------------------ binary16 */

float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: value = man * 2^-24 */
            float f = (float)man * 5.9604644775390625e-08f;
            memcpy(&bits, &f, 4);
            bits |= sign;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &bits, 4);
    return out;
}

uint16_t orc_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0));
    }
    if (absx >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (absx < 0x38800000u) { /* < 2^-14: subnormal or zero in binary16 */
        /* value * 2^24 as an integer with round-to-nearest-even */
        float a;
        memcpy(&a, &absx, 4);
        float scaled = a * 16777216.0f; /* exact: power-of-two scaling */
        float r = rintf(scaled);        /* default rounding mode: nearest even */
        return (uint16_t)(sign | (uint16_t)r);
    }
    uint32_t man = absx & 0x7fffffu;
    uint32_t exp = (absx >> 23) - 112;
    uint32_t h = (exp << 10) | (man >> 13);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h += 1; /* carries into exponent correctly */
    return (uint16_t)(sign | h);
}

/* ------------------------------------------------------------------ OpenCV INTER_CUBIC
 * Restated from OpenCV 3.4.x imgproc/resize.cpp (the reference pins opencv-python==3.4.5.20,
 * requirements.txt:58-59): interpolateCubic with A = -0.75; source coordinate
 * fx = (dx + 0.5) * scale - 0.5, sx = floor(fx); taps sx-1..sx+2 with indices clamped to the source
 * (border replicate); horizontal pass then vertical pass, float32, products summed left to right.
 * Parity UNPINNED: OpenCV is not installed here and the reference has no fixture for it. */

void orc_cubic_coeffs(float x, float c[4]) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_resize_cubic(const float *src, int sh, int sw, long s_ys, long s_xs, float *dst, int dh, int dw,
                      long d_ys, long d_xs, double scale_x, double scale_y) {
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    float *alpha = (float *)malloc(sizeof(float) * 4 * (size_t)dw);
    float *rows = (float *)malloc(sizeof(float) * 4 * (size_t)dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        xofs[dx] = sx;
        orc_cubic_coeffs(fx, alpha + 4 * dx);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        float beta[4];
        orc_cubic_coeffs(fy, beta);
        for (int k = 0; k < 4; k++) {
            int yy = clampi(sy - 1 + k, 0, sh - 1);
            const float *S = src + (long)yy * s_ys;
            float *D = rows + (size_t)k * dw;
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                const float *a = alpha + 4 * dx;
                float v = S[(long)clampi(sx - 1, 0, sw - 1) * s_xs] * a[0];
                v = v + S[(long)clampi(sx, 0, sw - 1) * s_xs] * a[1];
                v = v + S[(long)clampi(sx + 1, 0, sw - 1) * s_xs] * a[2];
                v = v + S[(long)clampi(sx + 2, 0, sw - 1) * s_xs] * a[3];
                D[dx] = v;
            }
        }
        float *out = dst + (long)dy * d_ys;
        for (int dx = 0; dx < dw; dx++) {
            float v = rows[dx] * beta[0];
            v = v + rows[(size_t)dw + dx] * beta[1];
            v = v + rows[(size_t)2 * dw + dx] * beta[2];
            v = v + rows[(size_t)3 * dw + dx] * beta[3];
            out[(long)dx * d_xs] = v;
        }
    }
    free(xofs);
    free(alpha);
    free(rows);
}

/* evaluate.py:77-80: cv2.resize(pafs, None, fx=4, fy=4, INTER_CUBIC) on the (h,w,30) array */
void orc_upsample4_planar_to_hwc(const float *src, int C, int h, int w, float *dst) {
    for (int c = 0; c < C; c++) {
        orc_resize_cubic(src + (size_t)c * h * w, h, w, w, 1, dst + c, 4 * h, 4 * w, (long)4 * w * C, C, 0.25, 0.25);
    }
}

/* ------------------------------------------------------------------ A2 flip-average
 * utils/parse_skeletons.py:80-103.  output[0] is the image, output[1] its mirror; the mirrored
 * sample is un-mirrored along W ([:, ::-1, :]) and its channels permuted ([:, :, flip_*_ord]),
 * then (a + b) / 2 element-wise IN THE ARRAY'S DTYPE (binary16 under AMP, parse_skeletons.py:75),
 * then .astype(float32) (:103). */

static inline float load_elem(const void *p, int is_f16, size_t i) {
    return is_f16 ? orc_f16_to_f32(((const uint16_t *)p)[i]) : ((const float *)p)[i];
}

static inline float avg2(float a, float b, int is_f16) {
    if (is_f16) {
        /* binary16 add then binary16 divide by 2, each correctly rounded (numpy half arithmetic:
         * computed in float32 and rounded back; neither step can double-round) */
        float s = orc_f16_to_f32(orc_f32_to_f16(a + b));
        return orc_f16_to_f32(orc_f32_to_f16(s / 2.0f));
    }
    return (a + b) / 2.0f;
}

void orc_flip_average(const void *net_out, int is_f16, int h, int w, int flip, float *heat, float *paf) {
    const size_t plane = (size_t)h * w;
    const size_t sample = (size_t)ORC_NUM_CH * plane;
    for (int c = 0; c < ORC_NUM_CH; c++) {
        float *dst;
        int src_c_flip;
        if (c < ORC_NUM_LIMB) {
            dst = paf + (size_t)c * plane;
            src_c_flip = FLIP_PAF_ORD[c];
        } else {
            dst = heat + (size_t)(c - ORC_NUM_LIMB) * plane;
            src_c_flip = ORC_NUM_LIMB + FLIP_HEAT_ORD[c - ORC_NUM_LIMB];
        }
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                float a = load_elem(net_out, is_f16, (size_t)c * plane + (size_t)y * w + x);
                if (flip) {
                    float b = load_elem(net_out, is_f16, sample + (size_t)src_c_flip * plane + (size_t)y * w + (w - 1 - x));
                    dst[(size_t)y * w + x] = avg2(a, b, is_f16);
                } else {
                    dst[(size_t)y * w + x] = a;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ A3 peak finding */

/* utils/parse_skeletons.py:115-119: maximum_filter(img, footprint=generate_binary_structure(2,1)) == img
 * AND img > thr.  scipy's default mode='reflect' duplicates the edge sample, so for a radius-1
 * window an out-of-bounds neighbour never exceeds the in-bounds ones: it is simply ignored.
 * np.nonzero order is row-major; the result is [x, y]. */
int orc_find_peaks_plus(const float *map, int h, int w, float thr, int *xy, int max_out) {
    int n = 0;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float v = map[(size_t)y * w + x];
            if (!(v > thr)) continue;
            float m = v;
            if (y > 0 && map[(size_t)(y - 1) * w + x] > m) m = map[(size_t)(y - 1) * w + x];
            if (y < h - 1 && map[(size_t)(y + 1) * w + x] > m) m = map[(size_t)(y + 1) * w + x];
            if (x > 0 && map[(size_t)y * w + x - 1] > m) m = map[(size_t)y * w + x - 1];
            if (x < w - 1 && map[(size_t)y * w + x + 1] > m) m = map[(size_t)y * w + x + 1];
            if (m == v) {
                if (n < max_out) {
                    xy[2 * n] = x;
                    xy[2 * n + 1] = y;
                }
                n++;
            }
        }
    }
    return n;
}

/* utils/util.py:177-185: reflect-pad 1 + 3x3 max-pool; keep hmax == heat AND heat >= thre.
 * torch 'reflect' mirrors without repeating the edge; for radius 1 that again only re-uses
 * in-bounds neighbours of the 3x3 window's own rows/cols -- but NOT the same set as ignoring
 * out-of-bounds: the padded value at x=-1 is map[.,1], which already lies inside the 3x3 window of
 * x=0.  So out-of-bounds neighbours are again ignored. */
int orc_find_peaks_3x3(const float *map, int h, int w, float thr, int *xy, int max_out) {
    int n = 0;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float v = map[(size_t)y * w + x];
            if (!(v >= thr)) continue;
            float m = v;
            for (int dy = -1; dy <= 1; dy++) {
                int yy = y + dy;
                if (yy < 0 || yy >= h) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    int xx = x + dx;
                    if (xx < 0 || xx >= w) continue;
                    float u = map[(size_t)yy * w + xx];
                    if (u > m) m = u;
                }
            }
            if (m == v) {
                if (n < max_out) {
                    xy[2 * n] = x;
                    xy[2 * n + 1] = y;
                }
                n++;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------ A4 heatmap_nms
 * utils/parse_skeletons.py:126-176.  Per peak: clip a (2*win+1)^2 window to the map (:143-144),
 * cv2.resize the PATCH by `upsample` with INTER_CUBIC (:149-153; borders replicate at the PATCH
 * edge, not the map edge), argmax (first occurrence, :156), and
 *   x = (px+0.5)*up-0.5 + col* - ((px-x_min+0.5)*up-0.5) = up*x_min + col*   (exact integers)
 * score = upsampled maximum (:163). */
int orc_heatmap_nms(const float *heat, int h, int w, int upsample, int refine, float *peaks_out, int max_out,
                    int part_count[ORC_NUM_PART]) {
    const int win = 2; /* parse_skeletons.py:135 */
    int total = 0;
    int cap = h * w;
    int *xy = (int *)malloc(sizeof(int) * 2 * (size_t)cap);
    float *patch_up = (float *)malloc(sizeof(float) * (size_t)((2 * win + 1) * upsample) * ((2 * win + 1) * upsample));
    for (int part = 0; part < ORC_NUM_PART; part++) {
        const float *map = heat + (size_t)part * h * w;
        int n = orc_find_peaks_plus(map, h, w, 0.1f, xy, cap); /* :139 hard-coded 0.1 */
        part_count[part] = n;
        for (int i = 0; i < n; i++) {
            int px = xy[2 * i], py = xy[2 * i + 1];
            float ox, oy, score;
            if (refine) {
                int x_min = px - win < 0 ? 0 : px - win;
                int y_min = py - win < 0 ? 0 : py - win;
                int x_max = px + win > w - 1 ? w - 1 : px + win;
                int y_max = py + win > h - 1 ? h - 1 : py + win;
                int pw = x_max - x_min + 1, ph = y_max - y_min + 1;
                int uw = pw * upsample, uh = ph * upsample; /* cv2: dsize = round(src * fx) */
                orc_resize_cubic(map + (size_t)y_min * w + x_min, ph, pw, w, 1, patch_up, uh, uw, uw, 1,
                                 1.0 / upsample, 1.0 / upsample);
                int best = 0;
                for (int k = 1; k < uw * uh; k++)
                    if (patch_up[k] > patch_up[best]) best = k; /* ndarray.argmax: first maximum */
                int row = best / uw, col = best % uw;
                ox = (float)(upsample * x_min + col);
                oy = (float)(upsample * y_min + row);
                score = patch_up[best];
            } else {
                /* :164-167 and :169-171 with refined_center = [0,0] */
                ox = ((float)px + 0.5f) * (float)upsample - 0.5f;
                oy = ((float)py + 0.5f) * (float)upsample - 0.5f;
                score = map[(size_t)py * w + px];
            }
            if (total < max_out) {
                float *r = peaks_out + (size_t)5 * total;
                r[0] = ox;
                r[1] = oy;
                r[2] = score;
                r[3] = (float)total; /* cnt_total_joints */
                r[4] = (float)part;  /* evaluate.py:99-103 appends joint_type */
            }
            total++;
        }
    }
    free(xy);
    free(patch_up);
    return total;
}

/* ------------------------------------------------------------------ A5-A7 process_paf */

typedef struct {
    int x, y;
    float score;
    int id;
} peak_t; /* pafprocess.h:29-34 */

typedef struct {
    int idx1, idx2;
    float score, overall_score, length;
    int gen; /* generation index: our tie-break, see below */
} cand_t; /* pafprocess.h:52-58 */

typedef struct {
    int id;      /* union {id, count}   pafprocess.h:36-45 */
    float score; /* union {score, length} */
} cpeak_t;

struct orc_ctx {
    peak_t *peaks_line;
    int n_peaks;
    cpeak_t *skeletons; /* n_skel x 20 */
    int n_skel, cap_skel;
    orc_connection *conns[ORC_NUM_LIMB];
    int n_conns[ORC_NUM_LIMB];
    int n_cands[ORC_NUM_LIMB];
    cand_t *cands[ORC_NUM_LIMB]; /* sorted accepted candidates, kept for stage-wise parity */
    int sort_oob;                /* the reference's sort would have read out of bounds */
};

orc_ctx *orc_create(void) { return (orc_ctx *)calloc(1, sizeof(orc_ctx)); }

static void ctx_reset(orc_ctx *c) {
    free(c->peaks_line);
    c->peaks_line = NULL;
    c->n_peaks = 0;
    free(c->skeletons);
    c->skeletons = NULL;
    c->n_skel = c->cap_skel = 0;
    for (int i = 0; i < ORC_NUM_LIMB; i++) {
        free(c->conns[i]);
        c->conns[i] = NULL;
        c->n_conns[i] = 0;
        c->n_cands[i] = 0;
        free(c->cands[i]);
        c->cands[i] = NULL;
    }
}

void orc_destroy(orc_ctx *c) {
    if (!c) return;
    ctx_reset(c);
    free(c);
}

/* pafprocess.cpp:329-331 */
static inline int round2int(float v) { return (int)(v + 0.5); }

/* ---- the reference's sort, restated --------------------------------------------------------
 * pafprocess.cpp:109 calls std::sort with comp_candidate (a.overall_score >= b.overall_score,
 * :333-335).  `>=` is not a strict weak ordering, and exactly tied candidates are COMMON on this
 * path: two adjacent NMS peaks often refine to the same up-sampled argmax (heatmap_nms), giving
 * duplicate peaks and hence bit-identical candidates.  Which duplicate wins the greedy pick decides
 * peak ids and, downstream, how skeletons merge.  So the order among ties is part of the result,
 * and it is whatever libstdc++'s introsort does.  Below is that algorithm (GCC libstdc++
 * bits/stl_algo.h __sort/__introsort_loop/__unguarded_partition/__final_insertion_sort and
 * bits/stl_heap.h, threshold 16, depth limit 2*floor(log2 n)) with the same comparator.
 *
 * libstdc++'s unguarded scans assume a strict weak ordering; with `>=` and ties at an extreme they
 * run off the array (undefined behaviour in the reference: it reads neighbouring heap memory and
 * may crash).  We stop such a scan at the array bound and raise `sort_oob`; for those inputs the
 * reference has no defined result and tests skip the comparison against the compiled reference. */
static int sort_oob_flag;

static inline int comp_ge(const cand_t *a, const cand_t *b) { return a->overall_score >= b->overall_score; }

static inline void cswap(cand_t *a, cand_t *b) {
    cand_t t = *a;
    *a = *b;
    *b = t;
}

static void ls_unguarded_linear_insert(cand_t *base, int last) {
    cand_t val = base[last];
    int next = last - 1;
    while (1) {
        if (next < 0) { /* the reference would read before the array */
            sort_oob_flag = 1;
            break;
        }
        if (!comp_ge(&val, &base[next])) break;
        base[last] = base[next];
        last = next;
        --next;
    }
    base[last] = val;
}

static void ls_insertion_sort(cand_t *base, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (comp_ge(&base[i], &base[first])) {
            cand_t val = base[i];
            memmove(&base[first + 1], &base[first], sizeof(cand_t) * (size_t)(i - first));
            base[first] = val;
        } else {
            ls_unguarded_linear_insert(base, i);
        }
    }
}

static void ls_push_heap(cand_t *b, int first, int hole, int top, cand_t value) {
    int parent = (hole - 1) / 2;
    while (hole > top && comp_ge(&b[first + parent], &value)) {
        b[first + hole] = b[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    b[first + hole] = value;
}

static void ls_adjust_heap(cand_t *b, int first, int hole, int len, cand_t value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp_ge(&b[first + child], &b[first + child - 1])) child--;
        b[first + hole] = b[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        b[first + hole] = b[first + child - 1];
        hole = child - 1;
    }
    ls_push_heap(b, first, hole, top, value);
}

static void ls_heapsort(cand_t *b, int first, int last) { /* __partial_sort(first, last, last) */
    int len = last - first;
    if (len >= 2) { /* __make_heap */
        int parent = (len - 2) / 2;
        while (1) {
            cand_t v = b[first + parent];
            ls_adjust_heap(b, first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) { /* __sort_heap / __pop_heap */
        --last;
        cand_t v = b[last];
        b[last] = b[first];
        ls_adjust_heap(b, first, 0, last - first, v);
    }
}

static void ls_move_median_to_first(cand_t *b, int result, int a, int bb, int c) {
    if (comp_ge(&b[a], &b[bb])) {
        if (comp_ge(&b[bb], &b[c])) cswap(&b[result], &b[bb]);
        else if (comp_ge(&b[a], &b[c])) cswap(&b[result], &b[c]);
        else cswap(&b[result], &b[a]);
    } else if (comp_ge(&b[a], &b[c])) cswap(&b[result], &b[a]);
    else if (comp_ge(&b[bb], &b[c])) cswap(&b[result], &b[c]);
    else cswap(&b[result], &b[bb]);
}

static int ls_unguarded_partition(cand_t *b, int first, int last, int pivot, int n) {
    while (1) {
        while (1) {
            if (first >= n) { /* the reference would read past the array */
                sort_oob_flag = 1;
                break;
            }
            if (!comp_ge(&b[first], &b[pivot])) break;
            ++first;
        }
        --last;
        while (1) {
            if (last < 0) {
                sort_oob_flag = 1;
                break;
            }
            if (!comp_ge(&b[pivot], &b[last])) break;
            --last;
        }
        if (!(first < last)) return first;
        cswap(&b[first], &b[last]);
        ++first;
    }
}

static void ls_introsort_loop(cand_t *b, int first, int last, int depth_limit, int n) {
    while (last - first > 16) {
        if (depth_limit == 0) {
            ls_heapsort(b, first, last);
            return;
        }
        --depth_limit;
        int mid = first + (last - first) / 2;
        ls_move_median_to_first(b, first, first + 1, mid, last - 1);
        /* with the non-strict comparator the left scan may run past `last` and stop on an element of a neighbouring
         * range; libstdc++ then continues with [first, cut) -- reproduced as is (cut <= n: the scan stops at the array end) */
        int cut = ls_unguarded_partition(b, first + 1, last, first, n);
        ls_introsort_loop(b, cut, last, depth_limit, n);
        last = cut;
    }
}

static void libstdcxx_sort(cand_t *b, int n) {
    if (n <= 0) return;
    int lg = 0;
    while ((1 << (lg + 1)) <= n) lg++;
    ls_introsort_loop(b, 0, n, 2 * lg, n);
    if (n > 16) { /* __final_insertion_sort */
        ls_insertion_sort(b, 0, 16);
        for (int i = 16; i < n; ++i) ls_unguarded_linear_insert(b, i);
    } else {
        ls_insertion_sort(b, 0, n);
    }
}

static cpeak_t *skel_push(orc_ctx *c) {
    if (c->n_skel == c->cap_skel) {
        c->cap_skel = c->cap_skel ? 2 * c->cap_skel : 16;
        c->skeletons = (cpeak_t *)realloc(c->skeletons, sizeof(cpeak_t) * NUM_PART_OUTS * (size_t)c->cap_skel);
    }
    return c->skeletons + (size_t)NUM_PART_OUTS * c->n_skel++;
}

static void skel_erase(orc_ctx *c, int idx) {
    memmove(c->skeletons + (size_t)NUM_PART_OUTS * idx, c->skeletons + (size_t)NUM_PART_OUTS * (idx + 1),
            sizeof(cpeak_t) * NUM_PART_OUTS * (size_t)(c->n_skel - idx - 1));
    c->n_skel--;
}

static inline float fmaxf_(float a, float b) { return a < b ? b : a; } /* std::max */
static inline float fminf_(float a, float b) { return b < a ? b : a; } /* std::min */

int orc_process_paf(orc_ctx *c, int p1, int p2, int p3, const float *peaks, int f1, int f2, int f3,
                    const float *pafmap, int min_img_size) {
    (void)f1;
    ctx_reset(c);
    sort_oob_flag = 0;
    const int n_in = p1 * p2;

    /* pafprocess.cpp:29-41: bucket by part; ids follow INPUT order; x,y truncated to int */
    peak_t *bucket[ORC_NUM_PART];
    int nb[ORC_NUM_PART];
    for (int k = 0; k < ORC_NUM_PART; k++) {
        bucket[k] = (peak_t *)malloc(sizeof(peak_t) * (size_t)(n_in > 0 ? n_in : 1));
        nb[k] = 0;
    }
    int peak_cnt = 0;
    for (int i = 0; i < p1; i++) {
        for (int j = 0; j < p2; j++) {
            const float *r = peaks + (size_t)p3 * (j + (size_t)p2 * i);
            peak_t info;
            info.id = peak_cnt++;
            info.x = (int)r[0];
            info.y = (int)r[1];
            info.score = r[2];
            int part_id = (int)r[4];
            bucket[part_id][nb[part_id]++] = info;
        }
    }
    /* :43-48 flatten in part order */
    c->peaks_line = (peak_t *)malloc(sizeof(peak_t) * (size_t)(n_in > 0 ? n_in : 1));
    c->n_peaks = 0;
    for (int k = 0; k < ORC_NUM_PART; k++)
        for (int i = 0; i < nb[k]; i++) c->peaks_line[c->n_peaks++] = bucket[k][i];

    /* :51-130 candidate scoring + greedy matching per limb */
    for (int pair_id = 0; pair_id < ORC_NUM_LIMB; pair_id++) {
        const peak_t *la = bucket[LIMB_PAIRS[pair_id][0]];
        const peak_t *lb = bucket[LIMB_PAIRS[pair_id][1]];
        const int na = nb[LIMB_PAIRS[pair_id][0]], nbb = nb[LIMB_PAIRS[pair_id][1]];
        if (na == 0 && nbb == 0) continue;
        cand_t *cands = (cand_t *)malloc(sizeof(cand_t) * (size_t)(na * nbb > 0 ? na * nbb : 1));
        int nc = 0;
        for (int ia = 0; ia < na; ia++) {
            const peak_t *pa = &la[ia];
            for (int ib = 0; ib < nbb; ib++) {
                const peak_t *pb = &lb[ib];
                float vx = (float)(pb->x - pa->x), vy = (float)(pb->y - pa->y); /* :67-69 */
                float vec_length = sqrtf(vx * vx + vy * vy);                  /* :70 */
                if (vec_length < 1e-12) continue;                             /* :71 */
                int num_steps = round2int(vec_length + 1);                    /* :73 */
                if (num_steps > STEP_PAF) num_steps = STEP_PAF;
                /* get_paf_scores, :311-327 */
                const float step_x = (float)(pb->x - pa->x) / (float)(num_steps - 1);
                const float step_y = (float)(pb->y - pa->y) / (float)(num_steps - 1);
                float scores = 0.0f;
                int criterion1 = 0;
                for (int i = 0; i < num_steps; i++) {
                    int lx = round2int((float)pa->x + (float)i * step_x);
                    int ly = round2int((float)pa->y + (float)i * step_y);
                    float s = pafmap[pair_id + (size_t)f3 * (lx + (size_t)f2 * ly)]; /* PAF(y,x,k), :9 */
                    scores += s;                                                     /* :84-87 */
                    if (s > THRESH_PAF_SCORE) criterion1 += 1;
                }
                /* :92 -- float / int -> float; the min() term and the sum are double; stored to float */
                double prior = 0.5 * min_img_size / vec_length - 1.0;
                if (prior > 0.0) prior = 0.0;
                float criterion2 = (float)((double)(scores / (float)num_steps) + prior);
                float min_num_steps = (float)num_steps * THRESH_PAF_STEP_RATIO; /* :93 */
                if ((float)criterion1 > min_num_steps && criterion2 > 0) {      /* :95 */
                    cand_t cd;
                    cd.idx1 = ia;
                    cd.idx2 = ib;
                    cd.score = criterion2;
                    cd.overall_score =
                        PAF_OUT_WEIGHTS[0] * criterion2 + PAF_OUT_WEIGHTS[1] * pa->score + PAF_OUT_WEIGHTS[2] * pb->score;
                    cd.length = vec_length;
                    cd.gen = nc;
                    cands[nc++] = cd;
                }
            }
        }
        c->n_cands[pair_id] = nc;
        libstdcxx_sort(cands, nc); /* :109 */
        int max_connections = na < nbb ? na : nbb;          /* :111 */
        char *used1 = (char *)calloc((size_t)(na > 0 ? na : 1), 1);
        char *used2 = (char *)calloc((size_t)(nbb > 0 ? nbb : 1), 1);
        orc_connection *conns = (orc_connection *)malloc(sizeof(orc_connection) * (size_t)(max_connections > 0 ? max_connections : 1));
        int ncn = 0;
        for (int k = 0; k < nc; k++) { /* :113-129 */
            const cand_t *cd = &cands[k];
            if (!used1[cd->idx1] && !used2[cd->idx2]) {
                used1[cd->idx1] = 1;
                used2[cd->idx2] = 1;
                orc_connection cn;
                cn.peak_id1 = la[cd->idx1].id;
                cn.peak_id2 = lb[cd->idx2].id;
                cn.score = cd->score;
                cn.cid1 = cd->idx1;
                cn.cid2 = cd->idx2;
                cn.length = cd->length;
                conns[ncn++] = cn;
                if (ncn >= max_connections) break;
            }
        }
        c->conns[pair_id] = conns;
        c->n_conns[pair_id] = ncn;
        free(used1);
        free(used2);
        c->cands[pair_id] = cands;
    }

    /* :132-275 skeleton assembly.  NOTE: peak_infos_line is indexed by peak ID (:162 etc.), which
     * equals the flattened position only when the input is already grouped by part (it is, for the
     * joint_list evaluate.py builds).  We index the same way the reference does. */
    const peak_t *pl = c->peaks_line;
    for (int pair_id = 0; pair_id < ORC_NUM_LIMB; pair_id++) {
        const int part_id1 = LIMB_PAIRS[pair_id][0], part_id2 = LIMB_PAIRS[pair_id][1];
        for (int ci = 0; ci < c->n_conns[pair_id]; ci++) {
            const orc_connection cur = c->conns[pair_id][ci];
            int num_found = 0, idx1 = 0, idx2 = 0;
            for (int s = 0; s < c->n_skel; s++) { /* :143-150 */
                const cpeak_t *sk = c->skeletons + (size_t)NUM_PART_OUTS * s;
                if (sk[part_id1].id == cur.peak_id1 || sk[part_id2].id == cur.peak_id2) {
                    if (num_found == 0) idx1 = s;
                    if (num_found == 1) idx2 = s;
                    num_found += 1;
                }
            }
            if (num_found == 1) { /* :152-180 */
                cpeak_t *s1 = c->skeletons + (size_t)NUM_PART_OUTS * idx1;
                int min_len = (int)(s1[LIMB_INFO_IDX].score * (float)LIMB_LENGTH_RATE); /* :154 int truncation */
                if (s1[part_id2].id == NOT_ASSIGNED && (float)min_len > cur.length) {
                    s1[part_id2].id = cur.peak_id2;
                    s1[part_id2].score = cur.score;
                    s1[LIMB_INFO_IDX].id += 1;
                    s1[LIMB_INFO_IDX].score = fmaxf_(s1[LIMB_INFO_IDX].score, cur.length);
                    s1[SCORE_IDX].score += pl[cur.peak_id2].score + cur.score;
                } else if (s1[part_id2].id != cur.peak_id2 && s1[part_id2].score <= cur.score &&
                           (float)min_len > cur.length) {
                    /* :163-171 overwrite happens BEFORE the subtraction, so -= and += see the same operands */
                    s1[part_id2].id = cur.peak_id2;
                    s1[part_id2].score = cur.score;
                    s1[SCORE_IDX].score -= pl[s1[part_id2].id].score + s1[part_id2].score;
                    s1[SCORE_IDX].score += pl[cur.peak_id2].score + cur.score;
                    s1[LIMB_INFO_IDX].score = fmaxf_(s1[LIMB_INFO_IDX].score, cur.length);
                } else if (s1[part_id2].id == cur.peak_id2 && s1[part_id2].score <= cur.score) { /* :173-180 */
                    s1[part_id2].id = cur.peak_id2;
                    s1[part_id2].score = cur.score;
                    s1[SCORE_IDX].score -= pl[s1[part_id2].id].score + s1[part_id2].score;
                    s1[SCORE_IDX].score += pl[cur.peak_id2].score + cur.score;
                    s1[LIMB_INFO_IDX].score = fmaxf_(s1[LIMB_INFO_IDX].score, cur.length);
                }
            } else if (num_found == 2) { /* :182-256 */
                cpeak_t *s1 = c->skeletons + (size_t)NUM_PART_OUTS * idx1;
                cpeak_t *s2 = c->skeletons + (size_t)NUM_PART_OUTS * idx2;
                int min_len = (int)(s1[LIMB_INFO_IDX].score * (float)LIMB_LENGTH_RATE);
                int is_member = 0;
                float min1 = 0.0f, min2 = 0.0f;
                for (int kp = 0; kp < ORC_NUM_PART; kp++) {
                    int a1 = s1[kp].id > 0; /* :200-201: peak id 0 counts as unassigned */
                    int a2 = s2[kp].id > 0;
                    if (a1) min1 = (min1 == 0.0f) ? s1[kp].score : fminf_(min1, s1[kp].score);
                    if (a2) min2 = (min2 == 0.0f) ? s2[kp].score : fminf_(min2, s2[kp].score);
                    if (a1 && a2) is_member = 1;
                }
                if (!is_member) {
                    float lim = fminf_(min1, min2) * MIN_SCORE_TOLERANCE;
                    if (cur.score >= lim || cur.length < (float)min_len) { /* :221 OR, not AND */
                        for (int kp = 0; kp < ORC_NUM_PART; kp++) {
                            s1[kp].id += (s2[kp].id + 1);
                            s1[kp].score += (s2[kp].score + 1);
                        }
                        s1[LIMB_INFO_IDX].id += s2[LIMB_INFO_IDX].id;
                        s1[LIMB_INFO_IDX].score = fmaxf_(s1[LIMB_INFO_IDX].score, cur.length);
                        s1[SCORE_IDX].score += s2[SCORE_IDX].score + cur.score;
                        skel_erase(c, idx2);
                    }
                }
                /* else: :231-256 is gated by DELETE_SHARED_JOINTS == false -> dead */
            } else if (num_found == 0) { /* :257-273 */
                cpeak_t *ns = skel_push(c);
                for (int i = 0; i < NUM_PART_OUTS; i++) {
                    ns[i].id = NOT_ASSIGNED;
                    ns[i].score = (float)NOT_ASSIGNED;
                }
                ns[part_id1].id = cur.peak_id1;
                ns[part_id2].id = cur.peak_id2;
                ns[part_id1].score = cur.score;
                ns[part_id2].score = cur.score;
                ns[LIMB_INFO_IDX].id = 2;
                ns[LIMB_INFO_IDX].score = cur.length;
                ns[SCORE_IDX].score = pl[cur.peak_id1].score + pl[cur.peak_id2].score + cur.score;
            }
            /* num_found > 2: no action */
        }
    }

    /* :278-282 prune, from the back */
    for (int i = c->n_skel - 1; i >= 0; i--) {
        const cpeak_t *sk = c->skeletons + (size_t)NUM_PART_OUTS * i;
        if (sk[LIMB_INFO_IDX].id < THRESH_PART_CNT ||
            sk[SCORE_IDX].score / (float)sk[LIMB_INFO_IDX].id < THRESH_SKELETON_SCORE)
            skel_erase(c, i);
    }
    for (int k = 0; k < ORC_NUM_PART; k++) free(bucket[k]);
    c->sort_oob = sort_oob_flag;
    return 0; /* :284 */
}

int orc_get_num_humans(const orc_ctx *c) { return c->n_skel; }
int orc_get_part_peak_id(const orc_ctx *c, int s, int part) { return c->skeletons[(size_t)NUM_PART_OUTS * s + part].id; }
float orc_get_score(const orc_ctx *c, int s) {
    const cpeak_t *sk = c->skeletons + (size_t)NUM_PART_OUTS * s;
    return sk[SCORE_IDX].score / (float)sk[LIMB_INFO_IDX].id;
}
int orc_get_part_x(const orc_ctx *c, int cid) { return c->peaks_line[cid].x; }
int orc_get_part_y(const orc_ctx *c, int cid) { return c->peaks_line[cid].y; }
float orc_get_part_score(const orc_ctx *c, int cid) { return c->peaks_line[cid].score; }
int orc_get_num_connections(const orc_ctx *c, int limb) { return c->n_conns[limb]; }
void orc_get_connection(const orc_ctx *c, int limb, int i, orc_connection *out) { *out = c->conns[limb][i]; }
int orc_get_num_candidates(const orc_ctx *c, int limb) { return c->n_cands[limb]; }
int orc_get_num_peaks(const orc_ctx *c) { return c->n_peaks; }
int orc_get_sort_oob(const orc_ctx *c) { return c->sort_oob; }
void orc_get_candidate(const orc_ctx *c, int limb, int i, int *idx1, int *idx2, float *score, float *overall,
                       float *length) {
    const cand_t *cd = &c->cands[limb][i];
    *idx1 = cd->idx1;
    *idx2 = cd->idx2;
    *score = cd->score;
    *overall = cd->overall_score;
    *length = cd->length;
}

/* ------------------------------------------------------------------ whole path
 * evaluate.py:75-129 with --run_refactor --run_cpp: predict_refactor's flip-average, heatmap_nms with
 * stride 4, x4 bicubic upsample of the limb maps to (4h,4w,30) HWC, joint_list, process_paf. */
int orc_pipeline(orc_ctx *c, const void *net_out, int is_f16, int h, int w, int flip, int min_img_size,
                 float *peaks_out, int max_peaks, int *n_peaks_out) {
    const size_t plane = (size_t)h * w;
    float *heat = (float *)malloc(sizeof(float) * ORC_NUM_HEAT * plane);
    float *paf = (float *)malloc(sizeof(float) * ORC_NUM_LIMB * plane);
    orc_flip_average(net_out, is_f16, h, w, flip, heat, paf);
    int cap = (int)(ORC_NUM_PART * plane);
    float *peaks = (float *)malloc(sizeof(float) * 5 * (size_t)cap);
    int part_count[ORC_NUM_PART];
    int n = orc_heatmap_nms(heat, h, w, 4, 1, peaks, cap, part_count);
    if (n_peaks_out) *n_peaks_out = n;
    if (peaks_out) memcpy(peaks_out, peaks, sizeof(float) * 5 * (size_t)(n < max_peaks ? n : max_peaks));
    int nh = 0;
    if (n > 0) { /* evaluate.py:105 */
        float *paf_up = (float *)malloc(sizeof(float) * 16 * plane * ORC_NUM_LIMB);
        orc_upsample4_planar_to_hwc(paf, ORC_NUM_LIMB, h, w, paf_up);
        orc_process_paf(c, 1, n, 5, peaks, 4 * h, 4 * w, ORC_NUM_LIMB, paf_up, min_img_size);
        nh = c->n_skel;
        free(paf_up);
    } else {
        ctx_reset(c);
    }
    free(heat);
    free(paf);
    free(peaks);
    return nh;
}

/* ------------------------------------------------------------------ A10 refine_centroid
 * utils/util.py:188-213.  Border peaks are returned unrefined with the raw score; otherwise the
 * (2r+1)^2 box gives offset = sum(box * grid) / sum(box) and score = box.mean().
 * numpy: box is float32, grid is int64 -> products are float64; box.sum()/mean() are float32
 * reductions.  We accumulate in double and compare with a tolerance (tests state it). */
void orc_refine_centroid(const float *map, int h, int w, int x, int y, int radius, double out[3]) {
    int x_min = x - radius, x_max = x + radius + 1, y_min = y - radius, y_max = y + radius + 1;
    if (y_max > h || y_min < 0 || x_max > w || x_min < 0) {
        out[0] = x;
        out[1] = y;
        out[2] = map[(size_t)y * w + x];
        return;
    }
    double sx = 0, sy = 0, s = 0;
    for (int yy = y_min; yy < y_max; yy++)
        for (int xx = x_min; xx < x_max; xx++) {
            double v = map[(size_t)yy * w + xx];
            /* np.mgrid[-r:r+1, -r:r+1] -> x_grid varies along axis 0 (rows!), y_grid along axis 1.
             * The reference multiplies score_box[row, col] by x_grid[row, col] = row offset, so its
             * "offset_x" is really the ROW centroid (utils/util.py:206-208).  Restated as written. */
            sx += v * (double)(yy - y);
            sy += v * (double)(xx - x);
            s += v;
        }
    out[0] = x + sx / s;
    out[1] = y + sy / s;
    out[2] = s / (double)((2 * radius + 1) * (2 * radius + 1));
}

/* ------------------------------------------------------------------ A8: the pure-Python twins
 * find_connections (utils/parse_skeletons.py:324-410) + find_humans (:413-600) on the refactored path's inputs
 * (peaks from heatmap_nms, float32 up-sampled limb maps), restated with NumPy-2 scalar semantics (the golden vectors
 * of tests/golden were produced by importing the reference under NumPy 2.2; under the reference's own NumPy 1.16
 * `float32 + python float` promoted to float64 instead -- the differences are at the 1e-8 level and cannot be pinned
 * offline).  Rules that differ from the C++ path are marked  [!=cpp]. */

static float f32_pairwise_sum(const float *a, int n) { /* numpy's add.reduce on a contiguous float32 vector */
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    float r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

typedef struct {
    int i, j;
    double score, limb_len, overall;
    int gen;
} pycand_t;

static int pycand_cmp(const void *pa, const void *pb) { /* sorted(key=overall, reverse=True): stable */
    const pycand_t *a = (const pycand_t *)pa, *b = (const pycand_t *)pb;
    if (a->overall > b->overall) return -1;
    if (a->overall < b->overall) return 1;
    return a->gen < b->gen ? -1 : (a->gen > b->gen ? 1 : 0);
}

static double f64_pairwise_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* peaks: rows of 5 doubles [x, y, score, id, part].  paf32: (H, W, C) float32 (refactored path) or NULL;
 * paf64: planar (C, H, W) float64 (original path: predict's accumulators) or NULL. */
static int py_find_humans_core(const double *peaks, int n_peaks, const float *paf32, const double *paf64, int H, int W, int C,
                               int img_height, double *persons_out, int cap, int *n_conn_out) {
    /* all_peaks[k]: rows of part k in input order; joint_candidates = the same rows flattened in part order */
    int *part_idx[ORC_NUM_PART], part_n[ORC_NUM_PART];
    for (int k = 0; k < ORC_NUM_PART; k++) {
        part_idx[k] = (int *)malloc(sizeof(int) * (size_t)(n_peaks > 0 ? n_peaks : 1));
        part_n[k] = 0;
    }
    for (int i = 0; i < n_peaks; i++) {
        int part = (int)peaks[5 * i + 4];
        part_idx[part][part_n[part]++] = i;
    }
    double *cand = (double *)malloc(sizeof(double) * 4 * (size_t)(n_peaks > 0 ? n_peaks : 1)); /* joint_candidates */
    {
        int q = 0;
        for (int k = 0; k < ORC_NUM_PART; k++)
            for (int t = 0; t < part_n[k]; t++, q++)
                for (int e = 0; e < 4; e++) cand[4 * q + e] = peaks[5 * part_idx[k][t] + e];
    }
    /* ---- find_connections */
    typedef struct {
        double src_id, dst_id, score, i, j, limb_len;
    } conn6;
    conn6 *conns[ORC_NUM_LIMB];
    int nconn[ORC_NUM_LIMB], special[ORC_NUM_LIMB];
    for (int pair = 0; pair < ORC_NUM_LIMB; pair++) {
        const int ps = LIMB_PAIRS[pair][0], pd = LIMB_PAIRS[pair][1];
        const int ns = part_n[ps], nd = part_n[pd];
        conns[pair] = NULL;
        nconn[pair] = 0;
        special[pair] = (ns == 0 && nd == 0); /* :340-342 */
        if (special[pair]) continue;
        pycand_t *cs = (pycand_t *)malloc(sizeof(pycand_t) * (size_t)(ns * nd > 0 ? ns * nd : 1));
        int nc = 0;
        for (int i = 0; i < ns; i++) {
            const double *S = peaks + 5 * part_idx[ps][i];
            for (int j = 0; j < nd; j++) {
                const double *D = peaks + 5 * part_idx[pd][j];
                const double dx = D[0] - S[0], dy = D[1] - S[1];
                const double limb_len = sqrt(dx * dx + dy * dy);         /* :352 */
                long rn = lrint(limb_len + 1.0);                         /* round(): half to even (default mode) */
                int mid_num = rn < 20 ? (int)rn : 20;                    /* :353 */
                if (limb_len == 0.0) continue;                           /* :356 */
                float resp[20];
                double resp64[20];
                int cnt = 0;
                for (int t = 0; t < mid_num; t++) {                      /* np.round(np.linspace(...)) :361-362 [!=cpp] */
                    double lx, ly;
                    if (mid_num == 1) {
                        lx = S[0];
                        ly = S[1];
                    } else {
                        const double stepx = dx / (double)(mid_num - 1), stepy = dy / (double)(mid_num - 1);
                        lx = (stepx == 0.0) ? ((double)t / (double)(mid_num - 1)) * dx + S[0] : (double)t * stepx + S[0];
                        ly = (stepy == 0.0) ? ((double)t / (double)(mid_num - 1)) * dy + S[1] : (double)t * stepy + S[1];
                        if (t == mid_num - 1) {
                            lx = D[0];
                            ly = D[1];
                        }
                    }
                    const long ix = lrint(lx), iy = lrint(ly);
                    if (paf32) {
                        resp[t] = paf32[pair + (size_t)C * ((size_t)ix + (size_t)W * (size_t)iy)];
                        if (resp[t] > 0.1f) cnt++;                       /* thre2, :375 (float32 array > 0.1) */
                    } else {
                        resp64[t] = paf64[((size_t)pair * H + (size_t)iy) * W + (size_t)ix];
                        if (resp64[t] > 0.1) cnt++;
                    }
                }
                const double prior = 0.5 * (double)img_height / limb_len - 1.0;
                double connect_score; /* :366  min(prior, 0): python min keeps the np.float64 unless 0 < prior */
                int score_is_f32;
                if (paf32) {
                    const float mean32 = f32_pairwise_sum(resp, mid_num) / (float)mid_num;
                    if (0.0 < prior) {
                        connect_score = (double)mean32; /* float32 + int 0 -> float32 */
                        score_is_f32 = 1;
                    } else {
                        connect_score = (double)mean32 + prior; /* float32 + float64 -> float64 */
                        score_is_f32 = 0;
                    }
                } else {
                    const double mean64 = f64_pairwise_sum(resp64, mid_num) / (double)mid_num;
                    connect_score = 0.0 < prior ? mean64 : mean64 + prior;
                    score_is_f32 = 0;
                }
                const int criterion1 = (double)cnt > (double)mid_num * 0.8; /* :375 */
                const int criterion2 = connect_score > 0.0;
                if (criterion1 && criterion2) {
                    double half_cs = score_is_f32 ? (double)(0.5f * (float)connect_score) : 0.5 * connect_score;
                    pycand_t cd;
                    cd.i = i;
                    cd.j = j;
                    cd.score = connect_score;
                    cd.limb_len = limb_len;
                    cd.overall = (half_cs + 0.25 * S[2]) + 0.25 * D[2]; /* :381 */
                    cd.gen = nc;
                    cs[nc++] = cd;
                }
            }
        }
        qsort(cs, (size_t)nc, sizeof(pycand_t), pycand_cmp); /* :391 */
        const int max_conn = ns < nd ? ns : nd;
        conns[pair] = (conn6 *)malloc(sizeof(conn6) * (size_t)(max_conn > 0 ? max_conn : 1));
        for (int k = 0; k < nc; k++) { /* :397-407 */
            int used = 0;
            for (int q = 0; q < nconn[pair]; q++)
                if (conns[pair][q].i == (double)cs[k].i || conns[pair][q].j == (double)cs[k].j) used = 1;
            if (used) continue;
            if (nconn[pair] >= max_conn) break;
            conn6 c6;
            c6.src_id = peaks[5 * part_idx[ps][cs[k].i] + 3];
            c6.dst_id = peaks[5 * part_idx[pd][cs[k].j] + 3];
            c6.score = cs[k].score;
            c6.i = cs[k].i;
            c6.j = cs[k].j;
            c6.limb_len = cs[k].limb_len;
            conns[pair][nconn[pair]++] = c6;
            if (nconn[pair] >= max_conn) break;
        }
        free(cs);
    }
    if (n_conn_out)
        for (int p = 0; p < ORC_NUM_LIMB; p++) n_conn_out[p] = nconn[p];

    /* ---- find_humans: person rows (20, 2) float64: [0..17] = {peak id, limb score}, [18] = {total, -1}, [19] = {count, len} */
    int np_ = 0, cap_p = 64;
    double *P = (double *)malloc(sizeof(double) * 40 * (size_t)cap_p);
#define PR(p, k, e) P[(size_t)40 * (p) + 2 * (k) + (e)]
    for (int limb = 0; limb < ORC_NUM_LIMB; limb++) {
        if (special[limb]) continue;
        const int st = LIMB_PAIRS[limb][0], dt = LIMB_PAIRS[limb][1];
        for (int ci = 0; ci < nconn[limb]; ci++) {
            const conn6 L = conns[limb][ci];
            int idx[2], nf = 0;
            for (int p = 0; p < np_; p++)
                if (PR(p, st, 0) == L.src_id || PR(p, dt, 0) == L.dst_id) {
                    if (nf >= 2) continue; /* :447-449 third and later matches are ignored [!=cpp] */
                    idx[nf++] = p;
                }
            if (nf == 1) {
                const int p = idx[0];
                const double dpk = PR(p, dt, 0), dsc = PR(p, dt, 1), plen = PR(p, 19, 1);
                if ((int)dpk == -1 && plen * 16.0 > L.limb_len) { /* :458 float compare, no int truncation [!=cpp] */
                    PR(p, dt, 0) = L.dst_id;
                    PR(p, dt, 1) = L.score;
                    PR(p, 19, 0) += 1;
                    PR(p, 19, 1) = L.limb_len > plen ? L.limb_len : plen;
                    PR(p, 18, 0) += cand[4 * (int)L.dst_id + 2] + L.score;
                } else if ((int)dpk != (int)L.dst_id && dsc <= L.score && plen * 16.0 > L.limb_len) {
                    PR(p, 18, 0) -= cand[4 * (int)dpk + 2] + dsc; /* the OLD peak and score are subtracted [!=cpp] */
                    PR(p, dt, 0) = L.dst_id;
                    PR(p, dt, 1) = L.score;
                    PR(p, 19, 1) = L.limb_len > plen ? L.limb_len : plen;
                    PR(p, 18, 0) += cand[4 * (int)L.dst_id + 2] + L.score;
                } else if ((int)dpk == (int)L.dst_id && dsc <= L.score) {
                    PR(p, 18, 0) -= cand[4 * (int)dpk + 2] + dsc;
                    PR(p, dt, 0) = L.dst_id;
                    PR(p, dt, 1) = L.score;
                    PR(p, 19, 1) = L.limb_len > plen ? L.limb_len : plen;
                    PR(p, 18, 0) += cand[4 * (int)L.dst_id + 2] + L.score;
                }
            } else if (nf == 2) {
                const int p1 = idx[0], p2 = idx[1];
                const double plen = PR(p1, 19, 1);
                int shared = 0;
                double min1 = 0, min2 = 0;
                int have1 = 0, have2 = 0;
                for (int k = 0; k < ORC_NUM_PART; k++) {
                    const int m1 = PR(p1, k, 0) >= 0, m2 = PR(p2, k, 0) >= 0; /* :502-503 `>= 0` [!=cpp] */
                    if (m1 && m2) shared = 1;
                    if (m1 && (!have1 || PR(p1, k, 1) < min1)) { min1 = PR(p1, k, 1); have1 = 1; }
                    if (m2 && (!have2 || PR(p2, k, 1) < min2)) { min2 = PR(p2, k, 1); have2 = 1; }
                }
                if (!shared) {
                    const double mt = min1 < min2 ? min1 : min2;
                    if (L.score >= 0.7 * mt && L.limb_len < plen * 16.0) { /* :511-512 AND [!=cpp] */
                        for (int k = 0; k < ORC_NUM_PART; k++)
                            for (int e = 0; e < 2; e++)
                                if (PR(p2, k, e) > PR(p1, k, e)) PR(p1, k, e) = PR(p2, k, e); /* np.maximum :516 [!=cpp] */
                        PR(p1, 19, 0) += PR(p2, 19, 0);
                        PR(p1, 19, 1) = L.limb_len > plen ? L.limb_len : plen;
                        PR(p1, 18, 0) += PR(p2, 18, 0) + L.score;
                        memmove(&PR(p2, 0, 0), &PR(p2 + 1, 0, 0), sizeof(double) * 40 * (size_t)(np_ - p2 - 1));
                        np_--;
                    }
                }
            } else { /* nobody claims either joint: new person, :583-596 */
                if (np_ == cap_p) {
                    cap_p *= 2;
                    P = (double *)realloc(P, sizeof(double) * 40 * (size_t)cap_p);
                }
                for (int k = 0; k < 20; k++) PR(np_, k, 0) = PR(np_, k, 1) = -1.0;
                PR(np_, st, 0) = L.src_id;
                PR(np_, st, 1) = L.score;
                PR(np_, dt, 0) = L.dst_id;
                PR(np_, dt, 1) = L.score;
                PR(np_, 19, 0) = 2;
                PR(np_, 19, 1) = L.limb_len;
                PR(np_, 18, 0) = (cand[4 * (int)L.src_id + 2] + cand[4 * (int)L.dst_id + 2]) + L.score;
                np_++;
            }
        }
    }
    int n_out = 0;
    for (int p = 0; p < np_; p++) { /* :599-603 */
        if (PR(p, 19, 0) < 2 || PR(p, 18, 0) / PR(p, 19, 0) < 0.45) continue;
        if (n_out < cap) memcpy(persons_out + (size_t)40 * n_out, &PR(p, 0, 0), sizeof(double) * 40);
        n_out++;
    }
#undef PR
    free(P);
    free(cand);
    for (int k = 0; k < ORC_NUM_PART; k++) free(part_idx[k]);
    for (int p = 0; p < ORC_NUM_LIMB; p++) free(conns[p]);
    return n_out;
}


int orc_py_find_humans(const float *peaks, int n_peaks, const float *paf, int H, int W, int C, int img_height,
                       double *persons_out, int cap, int *n_conn_out) {
    double *pk = (double *)malloc(sizeof(double) * 5 * (size_t)(n_peaks > 0 ? n_peaks : 1));
    for (int i = 0; i < 5 * n_peaks; i++) pk[i] = (double)peaks[i];
    int n = py_find_humans_core(pk, n_peaks, paf, NULL, H, W, C, img_height, persons_out, cap, n_conn_out);
    free(pk);
    return n;
}

int orc_py_find_humans_f64(const double *peaks, int n_peaks, const double *paf_planar, int H, int W, int img_height,
                           double *persons_out, int cap, int *n_conn_out) {
    return py_find_humans_core(peaks, n_peaks, NULL, paf_planar, H, W, ORC_NUM_LIMB, img_height, persons_out, cap, n_conn_out);
}

/* ------------------------------------------------------------------ A10: the original (non-refactored) path
 * predict (utils/parse_skeletons.py:180-283): per scale, the flip-averaged maps are up-sampled x4 (bicubic, :252-263),
 * the padding is cropped (:272-273), the result is resized to the image size (:276-277) and averaged over the scales
 * in float64 accumulators (:280-281; `heatmap / n` is a float32 division).  Planar layout here. */
void orc_predict_accumulate(const void *net_out, int is_f16, int h, int w, int flip, int pad_down, int pad_right, int img_h,
                            int img_w, int n_scales, double *heat_acc, double *paf_acc) {
    const size_t plane = (size_t)h * w;
    float *heat = (float *)malloc(sizeof(float) * ORC_NUM_HEAT * plane);
    float *paf = (float *)malloc(sizeof(float) * ORC_NUM_LIMB * plane);
    orc_flip_average(net_out, is_f16, h, w, flip, heat, paf);
    const int uh = 4 * h, uw = 4 * w;
    const int ch = uh - pad_down, cw = uw - pad_right; /* size of the scaled (unpadded) image */
    float *up = (float *)malloc(sizeof(float) * (size_t)uh * uw);
    float *rs = (float *)malloc(sizeof(float) * (size_t)img_h * img_w);
    for (int c = 0; c < ORC_NUM_CH; c++) {
        const float *src = c < ORC_NUM_LIMB ? paf + (size_t)c * plane : heat + (size_t)(c - ORC_NUM_LIMB) * plane;
        double *acc = c < ORC_NUM_LIMB ? paf_acc + (size_t)c * img_h * img_w : heat_acc + (size_t)(c - ORC_NUM_LIMB) * img_h * img_w;
        orc_resize_cubic(src, h, w, w, 1, up, uh, uw, uw, 1, 0.25, 0.25);
        if (ch == img_h && cw == img_w) { /* cv2.resize returns a copy when the size does not change */
            for (int y = 0; y < img_h; y++) memcpy(rs + (size_t)y * img_w, up + (size_t)y * uw, sizeof(float) * (size_t)img_w);
        } else {
            /* cv2.resize(src, (img_w, img_h)): scale = 1 / (dsize / ssize) in double */
            orc_resize_cubic(up, ch, cw, uw, 1, rs, img_h, img_w, img_w, 1, 1.0 / ((double)img_w / (double)cw),
                             1.0 / ((double)img_h / (double)ch));
        }
        for (size_t i = 0; i < (size_t)img_h * img_w; i++) acc[i] = acc[i] + (double)(rs[i] / (float)n_scales);
    }
    free(heat);
    free(paf);
    free(up);
    free(rs);
}

/* find_peaks (utils/parse_skeletons.py:286-321): float32 cast, 3x3 / >= thre1 NMS, refine_centroid (radius 2),
 * ids sequential over parts.  rows: [x, y, score, id, part] as doubles (coordinates are fractional here). */
int orc_find_peaks_original(const double *heat_acc, int img_h, int img_w, float thre1, double *rows_out, int max_rows) {
    const size_t plane = (size_t)img_h * img_w;
    float *m = (float *)malloc(sizeof(float) * plane);
    int *xy = (int *)malloc(sizeof(int) * 2 * plane);
    int total = 0;
    for (int part = 0; part < ORC_NUM_PART; part++) {
        for (size_t i = 0; i < plane; i++) m[i] = (float)heat_acc[(size_t)part * plane + i];
        int n = orc_find_peaks_3x3(m, img_h, img_w, thre1, xy, (int)plane);
        for (int i = 0; i < n; i++) {
            double o[3];
            orc_refine_centroid(m, img_h, img_w, xy[2 * i], xy[2 * i + 1], 2, o);
            if (total < max_rows) {
                double *r = rows_out + (size_t)5 * total;
                r[0] = o[0];
                r[1] = o[1];
                r[2] = o[2];
                r[3] = (double)total;
                r[4] = (double)part;
            }
            total++;
        }
    }
    free(m);
    free(xy);
    return total;
}

/* cv2.resize(INTER_CUBIC) on an 8-bit image (predict :204 for scale != 1): OpenCV's fixed-point path restated --
 * coefficients scaled by 2048 and rounded to short, integer horizontal pass, vertical pass (sum + 2^21) >> 22 with
 * saturation (resize.cpp: HResizeCubic<uchar,int,short>, VResizeCubic + FixedPtCast<int,uchar,22>, scalar form; the
 * SSE form of the vertical pass rounds in float and can differ by one count).  PARITY UNPINNED (OpenCV absent). */
void orc_resize_cubic_u8(const unsigned char *src, int sh, int sw, int cn, unsigned char *dst, int dh, int dw,
                         double scale_x, double scale_y) {
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        float cb[4];
        orc_cubic_coeffs(fy, cb);
        int ib[4];
        for (int k = 0; k < 4; k++) ib[k] = (int)lrintf(cb[k] * 2048.0f);
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= (float)sx;
            float ca[4];
            orc_cubic_coeffs(fx, ca);
            int ia[4];
            for (int k = 0; k < 4; k++) ia[k] = (int)lrintf(ca[k] * 2048.0f);
            for (int c = 0; c < cn; c++) {
                int acc = 0;
                for (int ky = 0; ky < 4; ky++) {
                    const unsigned char *row = src + (size_t)clampi(sy - 1 + ky, 0, sh - 1) * sw * cn;
                    int hsum = 0;
                    for (int kx = 0; kx < 4; kx++) hsum += (int)row[(size_t)clampi(sx - 1 + kx, 0, sw - 1) * cn + c] * ia[kx];
                    acc += hsum * ib[ky];
                }
                int v = (acc + (1 << 21)) >> 22;
                dst[((size_t)dy * dw + dx) * cn + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}
