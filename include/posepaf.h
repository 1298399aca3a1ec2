/*
 * posepaf.h -- C ABI of libposepaf.so: the MI355X (gfx950) post-processing path of the bottom-up pose
 * pipeline (flip-average -> heat-map NMS/peak refinement -> limb-map line-integral scoring -> greedy
 * limb matching -> person assembly), hand-written HIP behind plain pointers and sizes.
 *
 * Two groups of entry points:
 *
 *  (1) DROP-IN for the reference's native module `utils/pafprocess` -- the seven functions of
 *      /root/reference/utils/pafprocess/pafprocess.h:70-76, same names, argument order and meaning,
 *      exported with C linkage.  The SWIG interface (utils/pafprocess/pafprocess.i:14) or a ctypes stub
 *      binds them unchanged; see INTEGRATION.md.  Host arrays in, results held until the next call.
 *
 *  (2) NATIVE batched path -- takes the network output where it already lives (HBM) and returns
 *      fixed-size per-image records; this is what removes the reference's D2H copy
 *      (utils/parse_skeletons.py:80), the per-peak cv2.resize loop (:143-163) and the 31.5 MB limb-map
 *      upsample (evaluate.py:77-80).
 *
 * All functions return 0 (PP_OK) or a negative pp_status unless stated otherwise.  No function falls back
 * to a CPU implementation: without a usable HIP device every compute entry point returns PP_ERR_NO_DEVICE.
 */
#ifndef POSEPAF_H
#define POSEPAF_H

#include <stdint.h>

/* exported symbols (the library is built with -fvisibility=hidden) */
#define PP_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

#define PP_NUM_PART 18    /* pafprocess.h:11  NUM_PART */
#define PP_NUM_LIMB 30    /* pafprocess.h:20  COCOPAIRS_SIZE */
#define PP_NUM_HEAT 20    /* utils/parse_skeletons.py:17 */
#define PP_NUM_CH 50      /* config/config.py:127-129: [0,30) limb maps, [30,48) keypoints, [48,50) background */
#define PP_MAX_HUMANS 128 /* capacity of one pp_record */
#define PP_MAX_PEAKS_PER_PART_LIMIT 128

typedef enum {
    PP_OK = 0,
    PP_ERR_NO_DEVICE = -1,   /* no HIP device / runtime error at create */
    PP_ERR_BAD_ARG = -2,
    PP_ERR_TOO_LARGE = -3,   /* batch/h/w beyond what the context was created for, or map does not fit LDS */
    PP_ERR_HIP = -4,         /* a HIP call failed; pp_last_hip_error() has the code */
    PP_ERR_OVERFLOW = -5,    /* compat path only: a per-part / per-image capacity was exceeded */
    PP_ERR_UNSUPPORTED = -6  /* pp_conv_f16: this tile configuration / channel count is not available for the shape */
} pp_status;

typedef enum { PP_F32 = 0, PP_F16 = 1 } pp_dtype;

/* per-image status bits in pp_record.status */
#define PP_ST_PEAK_OVERFLOW 1u   /* a keypoint channel had more peaks than max_peaks_per_part (extra dropped) */
#define PP_ST_HUMAN_OVERFLOW 2u  /* more than PP_MAX_HUMANS people after pruning (extra dropped) */
#define PP_ST_SKEL_OVERFLOW 4u   /* more live partial skeletons than the assembly table holds */
#define PP_ST_SORT_UNDEFINED 8u  /* the reference's std::sort (non-strict comparator, pafprocess.cpp:333-335)
                                    would have read outside its array on this input: its result is undefined */
#define PP_ST_CAND_OVERFLOW 16u  /* more accepted limb candidates for one limb than the kernel holds */
#define PP_ST_FLOAT_COORDS 32u   /* original path: pp_human.x / .y hold float32 BIT PATTERNS (fractional coordinates) */
#define PP_ST_SYNC_TIMEOUT 64u   /* the image's assembly gave up waiting for a limb of its own launch (cannot happen on a healthy
                                    device: the record is incomplete and the context must be re-created) */

/* One person.  Mirrors what evaluate.py:111-127 pulls through the getters:
 * peak_id[p] = get_part_peak_id(h,p) (-1 = part absent); x/y/part_score = get_part_x/y/score(peak_id);
 * score = get_score(h) = total_score / part_count (pafprocess.cpp:295-297). */
typedef struct {
    int32_t peak_id[PP_NUM_PART];
    int32_t x[PP_NUM_PART];
    int32_t y[PP_NUM_PART];
    float part_score[PP_NUM_PART];
    float score;
    int32_t n_parts;
} pp_human;

typedef struct {
    int32_t n_humans;
    int32_t n_peaks;
    uint32_t status;
    int32_t n_connections;
    pp_human humans[PP_MAX_HUMANS];
} pp_record;

typedef struct pp_ctx pp_ctx;

/* ---------------------------------------------------------------- context */

/* Allocates all device workspace up front (no allocation or synchronisation happens in the per-batch
 * calls, so they can be captured into a hipGraph).  max_h/max_w are FEATURE-map sizes (image/4).
 * max_peaks_per_part in [1, 128]; 64 covers every synthetic scene up to 45 people. */
PP_API int pp_create(pp_ctx **out, int device, int max_batch, int max_h, int max_w, int max_peaks_per_part);
PP_API int pp_destroy(pp_ctx *ctx);
PP_API int pp_last_hip_error(const pp_ctx *ctx);
PP_API const char *pp_status_string(int status);
/* 1 if a HIP device is visible to this process, 0 otherwise (does not create a context) */
PP_API int pp_device_available(void);

/* ---------------------------------------------------------------- native batched path
 * net_out_dev: DEVICE pointer, (batch, n_samples, 50, h, w) planar, n_samples = 2 when flip (sample 1 is the
 * network's output for the W-mirrored image, utils/parse_skeletons.py:69-72) else 1; dtype PP_F16 or PP_F32.
 * With PP_F16 the flip-average is done in binary16 and then widened, exactly like numpy on the float16
 * array of utils/parse_skeletons.py:91-93,103.
 * min_img_size: the `img_h` argument of process_paf (evaluate.py:110), one value for the whole batch;
 * min_img_size_dev (optional DEVICE int[batch]) overrides it per image.
 * records_dev: DEVICE pp_record[batch].  stream: hipStream_t (NULL = default stream).  Asynchronous. */
PP_API int pp_process_batch(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip,
                     int min_img_size, const int *min_img_size_dev, pp_record *records_dev, void *stream);

/* The same batched path with the rules of the reference's PURE-PYTHON matching, find_connections + find_humans
 * (utils/parse_skeletons.py:324-600) -- what evaluate.py runs with --run_refactor but WITHOUT --run_cpp.  It differs
 * from the C++ rules in sampling (np.round(np.linspace)), float64 arithmetic, a stable sort, the merge conditions and the
 * score bookkeeping.  img_height is find_connections' `img_height` argument.  Thresholds are the
 * INI defaults (utils/config:17-25).  Needs max_peaks_per_part <= 64. */
PP_API int pp_process_batch_py(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip,
                               int img_height, const int *img_height_dev, pp_record *records_dev, void *stream);

/* Config-2 path: flip-average + NMS + refinement of the 18 keypoint channels only
 * (utils/parse_skeletons.py:126-176).  peaks_dev: DEVICE float[batch][18][max_peaks_per_part][4] =
 * (x, y, score, unused); counts_dev: DEVICE int[batch][18] (true counts, may exceed the capacity).
 * refine = 0 reproduces bool_refine_center=False.  Either output may be NULL to use the context's own
 * workspace (read back with pp_read_peaks). */
PP_API int pp_nms_batch(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip, int refine,
                 float *peaks_dev, int *counts_dev, void *stream);

/* Same kernel with the original (non-refactored) path's rules selectable:
 *   nms_mode    0: plus-shaped window, value >  threshold   (find_peaks_refactor, utils/parse_skeletons.py:115-116)
 *               1: full 3x3 window,    value >= threshold   (util.keypoint_heatmap_nms, utils/util.py:177-185)
 *   refine_mode 0: (p + 0.5) * 4 - 0.5, raw score            (heatmap_nms with bool_refine_center=False)
 *               1: x4 bicubic patch arg-max                   (heatmap_nms, :143-163)
 *               2: 5x5 weighted centroid, score = box mean    (util.refine_centroid, utils/util.py:188-213)
 *               3: integer map coordinates, raw score */
PP_API int pp_nms_batch_ex(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip, int nms_mode,
                           float threshold, int refine_mode, float *peaks_dev, int *counts_dev, void *stream);

/* Measurement aid: runs the kernels of pp_process_batch `iters` times EACH on `stream`, bracketed by HIP
 * events on that stream, and returns the average duration of one launch in milliseconds:
 * ms_out[0] = k_heat_peaks (peaks + the image ordering), ms_out[1] = k_limb_connect (limb scoring, matching AND the
 * person assembly done by each image's last limb workgroup), ms_out[2] = k_assemble_wave (the assembly alone as its own
 * one-wave-per-image launch; diagnostic, not part of the chain), ms_out[3] = the whole chain as pp_process_batch enqueues it.
 * ms_out must hold 4 floats.  Blocking. */
PP_API int pp_time_kernels(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip,
                           int min_img_size, int iters, float *ms_out, void *stream);

/* Forward-pass helpers (not part of the reference's interface), all on channels-last (NHWC) fp16 DEVICE tensors with
 * channels % 8 == 0 and 16-byte aligned pointers:
 *   pp_bias_act_f16 : in place  y = act(y + bias[c] (+ residual)) (+ post),  act = LeakyReLU(slope) if has_act;
 *                     n_elems = N*H*W*C; bias fp16[channels]; residual / post may be NULL
 *   pp_maxpool2_f16 : y[n][ho][wo][c] = max of the 2x2 window of x (x is (n, 2*h_out, 2*w_out, c))
 *   pp_upsample2_f16: nearest x2: y (n, 2*h_in, 2*w_in, c) from x (n, h_in, w_in, c) */
PP_API int pp_bias_act_f16(void *y, const void *bias, const void *residual, const void *post, long n_elems, int channels,
                           float slope, int has_act, void *stream);
PP_API int pp_maxpool2_f16(const void *x, void *y, long n, int h_out, int w_out, int channels, void *stream);
PP_API int pp_upsample2_f16(const void *x, void *y, long n, int h_in, int w_in, int channels, void *stream);
/* y = a + b (+ c) on fp16 tensors of n_elems elements (multiple of 8, 16-byte aligned), fp32 sum rounded once; c may be NULL. */
PP_API int pp_add3_f16(const void *a, const void *b, const void *c, void *y, long n_elems, void *stream);
/* SE squeeze (models/layers_transposed.py SELayer, AdaptiveAvgPool2d(1)): out[n][c] = mean over the h*w pixels of the NHWC fp16
 * activation x (n, hw, channels), fp32 accumulation.  partial_ws: DEVICE float[n][splits][channels] scratch; splits = number of
 * workgroups per image (pick n*splits >= ~1024).  channels % 8 == 0, channels <= 2048. */
PP_API int pp_channel_mean_f16(const void *x, void *partial_ws, void *out, int n, long hw, int channels, int splits, void *stream);
/* The SE squeeze without a pass of its own (round 3): the 3x3 / pad 1 halo-tile kernel writes y = act(conv + bias) AND, per image,
 * per-split partial sums of its binary16 outputs: sums_ws DEVICE float[n][splits][c_out], splits = pp_conv_own_sums_splits(h, w)
 * (0: the shape is not taken by that kernel); pp_channel_mean_finish_f16 adds the partials in split order and writes the fp16 mean
 * (= pp_channel_mean_f16's second kernel). */
PP_API int pp_conv_own_sums_splits(int h, int w);
PP_API int pp_conv_own_sums_f16(const void *x, const void *w, const void *bias, void *y, void *sums_ws, int n, int h, int wd, int c_in,
                                int c_out, float slope, void *stream);
PP_API int pp_channel_mean_finish_f16(const void *partial_ws, void *out, int n, long hw, int channels, int splits, void *stream);
/* The SE block's excitation (models/layers_transposed.py:289-310) in one launch: gains (n, c) fp16 = sigmoid(W2 leaky(W1 mean + b1)
 * + b2), one workgroup per sample, roundings as the fp16 torch modules it replaces.  Input: partial_ws (n, splits, c) fp32 channel
 * sums over hw pixels (pp_conv_own_sums_f16 / the first pass of pp_channel_mean_f16) OR mean (n, c) fp16 -- the other NULL.
 * w1 (hidden, c), b1 (hidden), w2 (c, hidden), b2 (c): fp16 DEVICE. */
PP_API int pp_se_gains_f16(const void *partial_ws, const void *mean, const void *w1, const void *b1, const void *w2, const void *b2,
                           void *out, int n, long hw, int c, int hidden, int splits, float slope, void *stream);
/* The layout change between the forward and the post-processing: x DEVICE (n, hw, 64) fp16 -- the last head's pixel-major output,
 * 50 channels + zero padding -- to channel planes y DEVICE (n, c_out, hw), c_out <= 64: the (N, 50, h, w) tensor the reference's
 * network returns (models/posenet.py:193-202) and pp_process_batch reads.  One pass at streaming speed instead of a strided copy. */
PP_API int pp_nhwc64_to_planes_f16(const void *x, void *y, int n, long hw, int c_out, void *stream);
/* SE excitation: y[n][p][c] = x[n][p][c] * scale[n][c] on NHWC fp16 (x: (n, hw, channels), scale: fp16 (n, channels)); y may be x. */
PP_API int pp_channel_scale_f16(const void *x, const void *scale, void *y, int n, long hw, int channels, void *stream);

/* Fused point-wise (1x1) convolution on the matrix cores (v_mfma_f32_32x32x16_f16), fp16 in/out, fp32 accumulate:
 *   y[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n] (+ residual[m][n])) (+ post[m][n])
 * x: DEVICE [M][K] (channels-last activation, M = batch*H*W), w: DEVICE [N][K] (a (Cout, Cin, 1, 1) conv weight),
 * bias fp16[N], residual / post optional [M][N], y [M][N]; all 16-byte aligned.  pp_pwconv_supported(K, N) tells whether
 * the shape is taken (K in {64,128,192,256}, N % 32 == 0, weights fit LDS); other shapes stay on PyTorch-ROCm. */
PP_API int pp_pwconv_supported(int K, int N);
PP_API int pp_pwconv_f16(const void *x, const void *w, const void *bias, const void *residual, const void *post, void *y,
                         long M, int K, int N, float slope, int has_act, void *stream);

/* A1 forward: one fused convolution on the matrix cores, stride 1, square kernel, fp16 in/out, fp32 accumulate:
 *   y = leaky(conv(x, w) + bias[k] (+ extra if extra_mode == 1)) (+ extra if extra_mode == 2)
 * i.e. Conv2d + folded BatchNorm + LeakyReLU of models/layers_transposed.py:Conv, with the residual add of
 * Residual.forward (extra_mode 1) or the hourglass's `up1 +` (extra_mode 2) applied to the fp32 accumulators.
 * x: DEVICE (n, h, w, c_in) NHWC; w: DEVICE (c_out, ksize, ksize, c_in) (= a channels-last (c_out, c_in, k, k) weight);
 * bias fp16[c_out]; extra: DEVICE (n, ho, wo, c_out) or NULL (extra_mode 0); y: DEVICE (n, ho, wo, c_out) with
 * ho = h + 2*pad - dilation*(ksize-1).  slope = LeakyReLU slope (1.0f = no activation).  c_in, c_out multiples of 8,
 * all pointers 16-byte aligned.  config in [0, pp_conv_num_configs()) picks the workgroup tile (M x N x K-step per block):
 * 0 128x256x32, 1 256x128x32, 2 128x128x32, 3 128x64x32 (128 threads), 4 128x64x32, 5 64x128x32, and the software-pipelined
 * variants 6 128x128x64 (v4), 7 128x128x64 (v3), 8 256x256x32 (v3), 9 256x256x64 (v3);
 * callers time the configurations once per layer shape and keep the fastest (posepaf/fused_model.py).
 * The GEMM main loop is ROCm composable_kernel's XDL implicit-GEMM template; the epilogue functor is this library's. */
PP_API int pp_conv_num_configs(void);
PP_API int pp_conv_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd,
                       int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int config,
                       void *stream);

/* pp_conv_f16 with explicit pixel strides (elements, multiples of 8): x and / or y may be a channel slice of a wider NHWC tensor
 * (ldx >= c_in, ldy >= c_out) -- the two halves of the backbone's concatenation (models/layers_transposed.py:193-195) are
 * written in place by their producers and read in place by the dilated chain. */
PP_API int pp_conv_ld_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd,
                          int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int config, int ldx,
                          int ldy, void *stream);

/* A1 forward: the same fused convolution as pp_conv_f16 -- identical arguments and semantics -- as a HAND-WRITTEN implicit-GEMM
 * kernel (csrc/posepaf_conv_own.hip: 256-pixel x bn-channel workgroup tiles, LDS-DMA staging, v_mfma_f32_16x16x32_f16, epilogue
 * from registers; no composable_kernel).  Needs c_in % 32 == 0 and c_out % 64 == 0 (pp_conv_own_supported).  bn = output
 * channels per workgroup: 256, 128 or 64 (must divide c_out), 0 = the largest that divides c_out. */
PP_API int pp_conv_own_supported(int c_in, int c_out, int ksize);
/* Diagnostics (the diagnostics build only, `make -C csrc diag`: the product library compiles no ablated instance and ignores
 * POSEPAF_CONV_DBG): after a launch of the 3x3 halo kernel with POSEPAF_CONV_DBG bit 1024 set, the median over the first nwg
 * workgroups of the shader-clock cycles spent per section (summed over the workgroup's tiles): out6[0..4] = wait for the
 * first DMA and the previous stores / first fragment reads / main loop / next tile's decode + DMA issue / epilogue;
 * out6[5] = the main loop in 100 MHz ticks.  No reference counterpart. */
PP_API int pp_conv_debug_clock(double *out6, int nwg);
PP_API int pp_conv_own_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd,
                           int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int bn,
                           void *stream);
/* The same on channel SLICES of wider NHWC tensors (ldx / ldy = elements between consecutive pixels of x / y, multiples of 8; `extra`
 * stays packed) -- the backbone's concatenation written and read in place (models/layers_transposed.py:193-195).  The 3x3
 * halo-tile kernel only (bn = 512): pad = dilation = 1, or the backbone's dilated convolutions pad = dilation = 3 / 4 / 5
 * (models/layers_transposed.py:125-157, :175-182), whose rows the kernel walks class by class (y mod dilation). */
PP_API int pp_conv_own_ld_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd,
                              int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int bn, int ldx,
                              int ldy, void *stream);
/* The same with extensions.  For the 3x3 / pad 1 halo-tile kernel (bn = 512 only, PP_ERR_UNSUPPORTED otherwise):
 *  - upsampled_input = 1: x is (n, h/2, wd/2, c_in) and stands for its x2 nearest-neighbour upsample (h, wd even) -- the
 *    `self.upsample(down3)` of models/layers_transposed.py:272 (nn.Upsample, :212) without materialising the upsample;
 *  - extra_mode = 3: y = fp16(act(conv + bias)) + extra + extra2 (fp32 sum, rounded once) -- the hourglass' `up1 + up2`
 *    and the stage's feature-cache add (models/posenet.py:104-106) inside the convolution's epilogue.
 * For every kernel (bn = 0 / 256 / 128 / 64 / 512):
 *  - extra_mode = 4: y = act(conv + bias + extra) as in mode 1 and a SECOND OUTPUT y2 = y + extra2 (the sum of the two
 *    binary16 tensors, rounded once) -- `cache = merge_preds(...) + merge_features(...)` and `x = x + cache`
 *    (models/posenet.py:116-118) in one pass.  y2 is NULL in every other mode. */
PP_API int pp_conv_own_ex_f16(const void *x, const void *w, const void *bias, const void *extra, const void *extra2, void *y, void *y2,
                              int n, int h, int wd, int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope,
                              int bn, int upsampled_input, void *stream);

/* A1, the hourglass' `hg[i][3](upsample(low))` (models/layers_transposed.py:270-275): a 3x3 / pad 1 convolution behind a x2
 * nearest upsample, evaluated as FOUR 2x2 convolutions of the half-resolution tensor (one per output phase): an output pixel
 * (2y + py, 2x + px) only sees a 2x2 neighbourhood of the input, each input pixel through the sum of the 3x3 taps landing on it --
 * 16 instead of 36 multiply-adds per input pixel, the same real-number result.  w4: DEVICE [4][c_out][2][2][c_in] fp16, phase
 * (py, px) = (0,0), (0,1), (1,0), (1,1); rows: py = 0 -> {w[0], w[1] + w[2]} on input rows {y - 1, y}; py = 1 -> {w[0] + w[1], w[2]}
 * on {y, y + 1}; columns alike (sums formed in fp32, rounded once).  x: (n, h_low, w_low, c_in); y / extra / extra2: (n, 2 h_low,
 * 2 w_low, c_out); extra_mode 0 / 2 / 3 as pp_conv_own_ex_f16; bn: output channels per workgroup (0, 256, 128, 64). */
PP_API int pp_conv_up2_collapsed_f16(const void *x, const void *w4, const void *bias, const void *extra, const void *extra2, void *y, int n,
                                     int h_low, int w_low, int c_in, int c_out, int extra_mode, float slope, int bn, void *stream);

/* A1, the 1x1 convolutions as an HBM-bound stream with the neighbouring element-wise passes folded in (csrc/posepaf_conv_own.hip
 * k_pw):   y = act(conv1x1(x * scale[n]) + bias (+ extra))   [and  y2 = y + extra2]
 * x: DEVICE (m, c_in) fp16 = an NHWC activation of m = n * h * w pixels; scale: DEVICE (m / hw, c_in) fp16 or NULL -- the SE
 * block's per-sample channel gains (models/layers_transposed.py:289-310), multiplied into the input in binary16 exactly as the
 * separate x * s pass would; w: DEVICE (c_out, c_in); bias fp16[c_out]; extra / extra2 / y2: DEVICE (m, c_out) or NULL;
 * y: DEVICE with ldy >= c_out elements between pixels (a channel slice of a wider tensor when larger); hw = h * w.
 * extra_mode 0 / 1 / 2 / 4 as pp_conv_own_ex_f16; 5 = the second output y2 = y + extra2 without a tensor added before the
 * activation (extra = NULL).  pp_pw_supported: c_in in {64, 128, 192, 256, 384, 448, 512, 640, 704} (the sums of a two-input call included), c_out % 64 == 0;
 * with `scale`, hw % 64 == 0 (a group of pixels must not straddle two images). */
PP_API int pp_pw_supported(int c_in, int c_out);
PP_API int pp_pw_f16(const void *x, const void *scale, const void *w, const void *bias, const void *extra, const void *extra2, void *y,
                     void *y2, long m, int hw, int c_in, int c_out, int ldy, int extra_mode, float slope, void *stream);
/* The same with the 2x2 / stride-2 max-pool of the tensor it produces (y; y2 in mode 4) as one more output -- the hourglass pools
 * exactly these tensors (`low = hg[i][1](pool(x))`, models/layers_transposed.py:262-266), so its pooling passes disappear.
 * pool_out: DEVICE (n, h / 2, w / 2, c_out); width = w: a multiple of 32 (64 when c_in = 64); h even. */
PP_API int pp_pw_pool_f16(const void *x, const void *scale, const void *w, const void *bias, const void *extra, const void *extra2,
                          void *y, void *y2, void *pool_out, long m, int hw, int width, int c_in, int c_out, int ldy, int extra_mode,
                          float slope, void *stream);

/* pp_pw_f16 on TWO inputs whose channels continue each other along K: y = act(W [x ; x2] + bias (+ extra)), W: (c_out, c_in + c_in2).
 * A residual block's last 1x1 convolution and its 1x1 skip convolution (models/layers_transposed.py:12-48: conv3(t) + skip(x)) are one
 * product this way; the skip's output is never written.  pool_out (NULL: none) / width as pp_pw_pool_f16. */
PP_API int pp_pw_cat_f16(const void *x, const void *x2, const void *w, const void *bias, const void *extra, void *y, void *pool_out, long m,
                         int hw, int width, int c_in, int c_in2, int c_out, int ldy, int extra_mode, float slope, void *stream);

/* A1, the stem (models/layers_transposed.py:78-87 Backbone.conv1 + bn1 + LeakyReLU): y = leaky(conv(x, w, 7x7, stride 2, padding 3)
 * + bias) in one HBM-bound pass.  x: DEVICE (n, h, w, 3) NHWC fp16, h even, w % 4 == 0; w_prepared: DEVICE (64, 192) fp16,
 * w_prepared[k][(r * 8 + s1) * 3 + c] = weight[k][c][r][s1 - 1] (zeros for s1 = 0 and past 168); bias fp16[64];
 * y: DEVICE (n, h / 2, w / 2, 64) NHWC fp16. */
PP_API int pp_stem7x7_f16(const void *x, const void *w_prepared, const void *bias, void *y, int n, int h, int w, float slope,
                          void *stream);

/* A0 pre-processing (utils/parse_skeletons.py:52-73, utils/util.py:44-65) of a batch of equally sized BGR uint8 DEVICE
 * images (batch, h, w, 3): pad bottom/right to a multiple of pad_to with pad_value, divide by 255, and write each image
 * followed (flip != 0) by the W-mirror of the PADDED image.  out: DEVICE (batch*(flip?2:1), Hp, Wp, 3), PP_F16 or PP_F32. */
PP_API int pp_preprocess_u8(const void *images_u8, void *out, int dtype, int batch, int h, int w, int pad_to,
                            int pad_value, int flip, void *stream);
/* The same for a bucket of images of DIFFERENT sizes sharing one padded shape (hp, wp) -- the reference pads every image
 * alone (utils/parse_skeletons.py:54); the batched evaluation loop groups images by padded shape.  images_u8: DEVICE
 * (batch, hp, wp, 3), image b in the top-left (sizes[b], sizes[batch + b]) corner of its slot, the rest of the slot is never
 * read; sizes_dev: DEVICE int[2][batch] = heights, then widths (heights double as pp_process_batch's min_img_size_dev). */
PP_API int pp_preprocess_u8_ragged(const void *images_u8, const int *sizes_dev, void *out, int dtype, int batch, int hp,
                                   int wp, int pad_value, int flip, void *stream);

/* A2 standalone: the arrays predict_refactor returns (utils/parse_skeletons.py:82-103).  net_out_dev as for
 * pp_process_batch; heat_hwc_dev: DEVICE float[batch][h][w][20], paf_hwc_dev: DEVICE float[batch][h][w][30]. */
PP_API int pp_flip_average(const void *net_out_dev, int dtype, int batch, int h, int w, int flip, float *heat_hwc_dev,
                           float *paf_hwc_dev, void *stream);

/* Diagnostics: register a DEVICE buffer of 8 int64 per workgroup; K_A and K_B then store shader-clock stamps at
 * their phase boundaries (slot 0 start, 1 map in LDS, ...).  NULL (default) disables it. */
PP_API int pp_debug_set_stamps(long long *stamps_dev);
/* Diagnostics / A-B measurements of pp_process_batch's launch structure: 0 (default) two launches -- peaks, then limb matching
 * with the person assembly done by each image's last limb workgroup, images dispatched heaviest first; 1 = limb matching and
 * assembly as separate launches; 2 = as 0 without the load ordering.  Results are identical in every mode. */
PP_API int pp_debug_set_mode(pp_ctx *ctx, int mode);

/* Blocking read-backs of the context's workspace for the last batch (host pointers).
 * pp_read_peaks: joint_list rows [x, y, score, peak_id, part] (evaluate.py:99-103) of one image; returns
 * PP_ERR_OVERFLOW (rows still filled, truncated per part) when a part had more peaks than max_peaks_per_part. */
PP_API int pp_read_peaks(pp_ctx *ctx, int image, float *joint_list_host, int max_rows, int *n_rows);
/* connections of one limb of one image, rows {cid1, cid2, score, length} (pafprocess.h:60-67) */
PP_API int pp_read_connections(pp_ctx *ctx, int image, int limb, float *rows_host, int max_rows, int *n_rows);
/* number of peaks found in every keypoint channel of one image (counts[18]; may exceed max_peaks_per_part: the true count) */
PP_API int pp_read_part_counts(pp_ctx *ctx, int image, int *counts_host);
/* number of accepted connections of every limb of one image (counts[30]); valid after either rule set (C++ or Python twins) */
PP_API int pp_read_connection_counts(pp_ctx *ctx, int image, int *counts_host);
/* Diagnostics: the raw per-workgroup status words of one image, flags[48] = 18 part words (peak kernels) then 30 limb words
 * (limb kernels); pp_record.status is their OR plus the assembly's own flags.  Each word is plainly stored by its
 * workgroup on every launch; none is ever memset or accumulated. */
PP_API int pp_debug_read_flags(pp_ctx *ctx, int image, uint32_t *flags_host);
/* device -> host copy of `batch` records, then synchronises the stream */
PP_API int pp_read_records(pp_ctx *ctx, const pp_record *records_dev, pp_record *records_host, int batch, void *stream);

/* ---------------------------------------------------------------- drop-in, context form
 * Same contract as process_paf below but re-entrant: state lives in ctx. */
PP_API int pp_process_paf_host(pp_ctx *ctx, int p1, int p2, int p3, const float *peaks, int f1, int f2, int f3,
                        const float *pafmap, int min_img_size);
PP_API int pp_get_num_humans(const pp_ctx *ctx);
PP_API int pp_get_part_peak_id(const pp_ctx *ctx, int skeleton_id, int part_id);
PP_API float pp_get_score(const pp_ctx *ctx, int skeleton_id);
PP_API int pp_get_part_x(const pp_ctx *ctx, int cid);
PP_API int pp_get_part_y(const pp_ctx *ctx, int cid);
PP_API float pp_get_part_score(const pp_ctx *ctx, int cid);
PP_API uint32_t pp_get_status(const pp_ctx *ctx);

/* ---------------------------------------------------------------- original (non-refactored) path, SURVEY 8a row A10
 * predict (utils/parse_skeletons.py:180-283) + find_peaks (:286-321) + find_connections / find_humans at IMAGE
 * resolution, with a real scale search (BASELINE config 5).  All buffers are DEVICE memory owned by the caller:
 *   scratch_planar float[batch][50][h][w], scratch_up float[batch][50][4h][4w],
 *   heat_acc double[batch][20][img_h][img_w], paf_acc double[batch][30][img_h][img_w]  (zero them before the first scale),
 *   mask_scratch uchar[batch][18][img_h][img_w], peaks64_scratch 32 bytes x batch x 18 x max_peaks_per_part.
 * pp_original_accumulate adds ONE scale: flip-average, x4 bicubic, crop of (pad_down, pad_right) pixels, bicubic resize to
 * (img_h, img_w), += value / n_scales.  pp_original_finish: 3x3 / >= thre1 NMS + refine_centroid, Python-twin matching
 * on the float64 limb maps, records with PP_ST_FLOAT_COORDS (x / y are float32 bit patterns).
 * pp_resize_u8_cubic: cv2.resize(INTER_CUBIC) of a batch of interleaved 3-channel uint8 images (scale = src / dst). */
PP_API int pp_original_accumulate(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip,
                                  int pad_down, int pad_right, int img_h, int img_w, int n_scales, float *scratch_planar,
                                  float *scratch_up, double *heat_acc, double *paf_acc, void *stream);
/* All scales of predict's loop in ONE launch: the accumulators are WRITTEN (not added to; no zeroing needed) with
 * ((0 + v_1 / n) + v_2 / n) + ... in float64, bit-identical to n_scales calls of pp_original_accumulate on zeroed accumulators,
 * without the scratch maps and without re-reading the accumulators per scale.  net_out_dev[i]: DEVICE (batch, 2|1, 50, h[i],
 * w[i]); h / w / pad_down / pad_right: HOST arrays of n_scales entries (n_scales <= 6).  PP_ERR_UNSUPPORTED when a scale is so
 * large that its tiles do not fit LDS (more than ~3x the image size): use the per-scale form then. */
PP_API int pp_original_accumulate_all(pp_ctx *ctx, int batch, int n_scales, const void *const *net_out_dev, int dtype, const int *h,
                                      const int *w, int flip, const int *pad_down, const int *pad_right, int img_h, int img_w,
                                      double *heat_acc, double *paf_acc, void *stream);
PP_API int pp_original_finish(pp_ctx *ctx, int batch, int img_h, int img_w, float thre1, const double *heat_acc,
                              const double *paf_acc, unsigned char *mask_scratch, void *peaks64_scratch, pp_record *records_dev,
                              void *stream);
PP_API int pp_resize_u8_cubic(const void *src, void *dst, int batch, int sh, int sw, int dh, int dw, double scale_x,
                              double scale_y, void *stream);

/* ---------------------------------------------------------------- Python twins, host form
 * utils.parse_skeletons.find_connections / find_humans (:324-600) with host arrays, for callers of the non --run_cpp
 * branch of evaluate.py (:88-89).  peaks: joint-list rows [x, y, score, id, part] (n of them); paf: (H, W, C) float32.
 * conns: double[30][max_peaks_per_part][6] rows {src_id, dst_id, score, i, j, limb_len}; counts: int[30];
 * special: int[30] (1 where neither part has a peak); persons: rows of 40 doubles = person_to_joint_assoc (20, 2). */
PP_API int pp_py_find_connections_host(pp_ctx *ctx, const float *peaks, int n, const float *paf, int H, int W, int C,
                                       int img_height, double *conns_out, int *counts_out, int *special_out);
PP_API int pp_py_find_humans_host(pp_ctx *ctx, const double *conns, const int *counts, const float *peaks, int n,
                                  double *persons_out, int cap, int *n_out);

/* ---------------------------------------------------------------- drop-in, reference names
 * Replaces /root/reference/utils/pafprocess/pafprocess.h:70-76 one for one.
 *   peaks  : host float[p1][p2][p3], rows [x, y, score, peak_id, part]   (evaluate.py:99-107, p3 = 5)
 *   pafmap : host float[f1][f2][f3] = (H, W, 30) up-sampled limb maps     (evaluate.py:77-80, :109)
 * Results stay in one process-wide context until the next call (the reference keeps them in file-scope
 * globals, pafprocess.cpp:16-17); like the reference this group is not re-entrant.
 * process_paf returns 0 on success like the reference (pafprocess.cpp:284); unlike the reference it can
 * fail, and then returns a negative pp_status (no device, capacity exceeded) instead of computing on the CPU. */
PP_API int process_paf(int p1, int p2, int p3, float *peaks, int f1, int f2, int f3, float *pafmap, int min_img_size);
PP_API int get_num_humans(void);
PP_API int get_part_peak_id(int skeleton_id, int part_id);
PP_API float get_score(int skeleton_id);
PP_API int get_part_x(int cid);
PP_API int get_part_y(int cid);
PP_API float get_part_score(int cid);

#ifdef __cplusplus
}
#endif
#endif /* POSEPAF_H */
