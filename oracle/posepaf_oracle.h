/*
 * posepaf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's bottom-up pose post-processing
 * hot path.  It exists to CHECK the HIP path; nothing that ships may call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Pinning: process_paf/getters are checked against the reference's own C++
 * (compiled from /root/reference into oracle/_ref by oracle/Makefile) and against
 * tests/golden/ vectors produced by importing the reference's Python.  The two
 * OpenCV-dependent steps (cv2.resize INTER_CUBIC in heatmap_nms and the x4 PAF
 * upsample) are restated from OpenCV's published algorithm; OpenCV is absent in
 * the build container and the reference holds no fixture for them, so for those
 * two steps parity is UNPINNED.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef POSEPAF_ORACLE_H
#define POSEPAF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_PART 18      /* utils/pafprocess/pafprocess.h:11 */
#define ORC_NUM_LIMB 30      /* utils/pafprocess/pafprocess.h:20 */
#define ORC_NUM_HEAT 20      /* utils/parse_skeletons.py:17 */
#define ORC_NUM_CH   50      /* config/config.py:127-129 */

typedef struct orc_ctx orc_ctx;

typedef struct {
    int cid1, cid2;          /* index inside part A / part B peak lists */
    float score;             /* criterion2 */
    int peak_id1, peak_id2;  /* global peak ids */
    float length;
} orc_connection;            /* utils/pafprocess/pafprocess.h:60-67 */

/* ---- tables (config/config.py:114-121,150-152; pafprocess.h:21-27) ---- */
const int *orc_limb_pairs(void);     /* [30][2] */
const int *orc_flip_heat_ord(void);  /* [20] */
const int *orc_flip_paf_ord(void);   /* [30] */

/* ---- OpenCV INTER_CUBIC restatement (parity unpinned) ---- */
void orc_cubic_coeffs(float x, float c[4]);
/* src: sh x sw with element strides (s_ys, s_xs); dst: dh x dw with strides (d_ys, d_xs).
 * scale_x = src/dst ratio as cv2 computes it (1/fx when fx is given). */
void orc_resize_cubic(const float *src, int sh, int sw, long s_ys, long s_xs,
                      float *dst, int dh, int dw, long d_ys, long d_xs,
                      double scale_x, double scale_y);
/* planar (C,h,w) -> HWC (4h,4w,C), what evaluate.py:77-80 feeds process_paf */
void orc_upsample4_planar_to_hwc(const float *src, int C, int h, int w, float *dst);

/* ---- A2: flip-average, utils/parse_skeletons.py:80-103 ----
 * net_out: (2,50,h,w) planar.  is_f16: elements are IEEE binary16 (arithmetic done in binary16,
 * as numpy does on a float16 array), else float32.  Outputs planar float32:
 * heat (20,h,w) and paf (30,h,w).  flip=0 takes sample 0 only. */
void orc_flip_average(const void *net_out, int is_f16, int h, int w, int flip,
                      float *heat, float *paf);

/* ---- A3: find_peaks_refactor, utils/parse_skeletons.py:106-119 ----
 * plus-shaped 5-point max filter (scipy reflect == ignore out-of-bounds), strict > thr.
 * Writes (x,y) pairs row-major; returns count (writes at most max_out). */
int orc_find_peaks_plus(const float *map, int h, int w, float thr, int *xy, int max_out);

/* mode B: 3x3 window, >= thr (utils/util.py:177-185) */
int orc_find_peaks_3x3(const float *map, int h, int w, float thr, int *xy, int max_out);

/* ---- A4: heatmap_nms, utils/parse_skeletons.py:126-176 ----
 * heat: planar (>=18,h,w).  peaks_out: rows [x,y,score,peak_id,part] (the joint_list of
 * evaluate.py:99-103), ordered by part then row-major.  refine=0 reproduces
 * bool_refine_center=False.  Returns N (writes at most max_out rows). */
int orc_heatmap_nms(const float *heat, int h, int w, int upsample, int refine,
                    float *peaks_out, int max_out, int part_count[ORC_NUM_PART]);

/* ---- A5-A7: process_paf + getters, utils/pafprocess/pafprocess.cpp:26-309 ---- */
orc_ctx *orc_create(void);
void orc_destroy(orc_ctx *c);
int orc_process_paf(orc_ctx *c, int p1, int p2, int p3, const float *peaks,
                    int f1, int f2, int f3, const float *pafmap, int min_img_size);
int orc_get_num_humans(const orc_ctx *c);
int orc_get_part_peak_id(const orc_ctx *c, int skeleton_id, int part_id);
float orc_get_score(const orc_ctx *c, int skeleton_id);
int orc_get_part_x(const orc_ctx *c, int cid);
int orc_get_part_y(const orc_ctx *c, int cid);
float orc_get_part_score(const orc_ctx *c, int cid);
/* intermediate state, for stage-wise parity of the HIP kernels */
int orc_get_num_connections(const orc_ctx *c, int limb);
void orc_get_connection(const orc_ctx *c, int limb, int i, orc_connection *out);
int orc_get_num_candidates(const orc_ctx *c, int limb);   /* accepted candidate pairs before greedy */
int orc_get_num_peaks(const orc_ctx *c);
/* 1 if the reference's std::sort (non-strict comparator) would have read outside the candidate array:
 * the reference result is then undefined */
int orc_get_sort_oob(const orc_ctx *c);
/* i-th accepted candidate of `limb` in sorted (descending overall_score) order */
void orc_get_candidate(const orc_ctx *c, int limb, int i, int *idx1, int *idx2, float *score, float *overall,
                       float *length);

/* ---- whole path: net_out (2,50,h,w) -> humans, as evaluate.py:75-129 with --run_refactor --run_cpp ----
 * Returns number of humans; fills ctx for the getters.  peaks_out (optional) receives the joint_list. */
int orc_pipeline(orc_ctx *c, const void *net_out, int is_f16, int h, int w, int flip,
                 int min_img_size, float *peaks_out, int max_peaks, int *n_peaks_out);

/* ---- A8: the pure-Python twins find_connections + find_humans (utils/parse_skeletons.py:324-600), float64, on the
 * refactored path's inputs.  peaks: joint_list rows [x,y,score,id,part]; paf: (H,W,C) float32.
 * persons_out: rows of 40 doubles = (20, 2) as in `person_to_joint_assoc`; returns the number of persons. */
int orc_py_find_humans(const float *peaks, int n_peaks, const float *paf, int H, int W, int C, int img_height,
                       double *persons_out, int cap, int *n_conn_out);

/* the same on the original path's inputs: peaks as rows of 5 doubles, limb maps planar (30, H, W) float64 */
int orc_py_find_humans_f64(const double *peaks, int n_peaks, const double *paf_planar, int H, int W, int img_height,
                           double *persons_out, int cap, int *n_conn_out);

/* ---- A10: predict's per-scale accumulation (utils/parse_skeletons.py:250-281), planar float64 accumulators
 * heat_acc (20, img_h, img_w) and paf_acc (30, img_h, img_w), and find_peaks (:286-321). */
void orc_predict_accumulate(const void *net_out, int is_f16, int h, int w, int flip, int pad_down, int pad_right, int img_h,
                            int img_w, int n_scales, double *heat_acc, double *paf_acc);
int orc_find_peaks_original(const double *heat_acc, int img_h, int img_w, float thre1, double *rows_out, int max_rows);

/* cv2.resize(INTER_CUBIC) of an 8-bit interleaved image (fixed-point path; parity unpinned) */
void orc_resize_cubic_u8(const unsigned char *src, int sh, int sw, int cn, unsigned char *dst, int dh, int dw,
                         double scale_x, double scale_y);

/* ---- A10: util.refine_centroid, utils/util.py:188-213 (float64 arithmetic like numpy on f32->f64?) ---- */
void orc_refine_centroid(const float *map, int h, int w, int x, int y, int radius, double out_xys[3]);

/* binary16 helpers (round-to-nearest-even), exposed for tests */
uint16_t orc_f32_to_f16(float f);
float orc_f16_to_f32(uint16_t h);

#ifdef __cplusplus
}
#endif
#endif
