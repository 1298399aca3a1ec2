// posepaf_capi.hip -- the C ABI of libposepaf.so (include/posepaf.h): context/workspace management, the native
// batched entry points and the drop-in replacements of the reference's utils/pafprocess functions.
// Everything numeric happens in the HIP kernels (posepaf_kernels.hip); the host code only validates arguments,
// groups the caller's peak rows by part (the reference's own bucketing loop, pafprocess.cpp:29-41, which is
// index bookkeeping) and moves bytes.  There is no CPU implementation to fall back to.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "posepaf_internal.h"

struct pp_ctx {
    int device = 0;
    int max_batch = 0, max_h = 0, max_w = 0, maxp = 0, cap = 0;
    int hip_err = 0;
    // device workspace
    float4 *d_peaks = nullptr;    // [max_batch][18][maxp] (x, y, score, id bits)
    int *d_counts = nullptr;      // [max_batch][18]
    float4 *d_conns = nullptr;    // [max_batch][30][maxp] (cid1 bits, cid2 bits, score, length)
    int *d_conn_counts = nullptr; // [max_batch][30]
    float4 *d_conn_aux = nullptr; // [max_batch][30][maxp] (peak id 1, peak id 2, peak score 1, peak score 2): K_B -> assembly
    int *d_sync = nullptr;        // [32 * max_batch + 16]: image order [max_batch] | per-image launch counters [max_batch] | K_A's
                                  // 16 words | per-limb publication flags [max_batch][30] (see k_limb_connect); then: K_A's
                                  // grid-wide ticket; zeroed ONCE here, re-armed by the kernels themselves (no memset node)
    int mode = 0;                 // pp_debug_set_mode
    unsigned *d_status = nullptr; // [max_batch][48] flag words, one per producing workgroup (posepaf_kernels.hip or_flags)
    void *d_conns_py = nullptr;   // [max_batch][30][maxp] double4 (src, dst, score, length): Python-twin path
    double *d_persons = nullptr;  // [128][40] raw person table of the Python-twin host form
    int *d_npersons = nullptr;
    pp_record *d_records = nullptr;  // [max_batch] (used when the caller passes NULL, and by the drop-in path)
    float *d_paf = nullptr;       // drop-in path: the caller's up-sampled (H,W,C) map
    size_t d_paf_bytes = 0;
    hipStream_t last_stream = nullptr;
    int last_batch = 0;
    const float4 *last_peaks = nullptr;
    const int *last_counts = nullptr;
    // drop-in path results (host)
    pp_record h_record;
    std::vector<int> line_x, line_y;
    std::vector<float> line_s;
    bool have_result = false;
};

namespace {

int fail_hip(pp_ctx *c, hipError_t e) {
    if (c) c->hip_err = (int)e;
    return PP_ERR_HIP;
}

#define PP_HIP(ctx, call)                               \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) return fail_hip(ctx, e__); \
    } while (0)

void free_ctx(pp_ctx *c) {
    if (!c) return;
    (void)hipFree(c->d_peaks);
    (void)hipFree(c->d_counts);
    (void)hipFree(c->d_conns);
    (void)hipFree(c->d_conn_counts);
    (void)hipFree(c->d_conn_aux);
    (void)hipFree(c->d_sync);
    (void)hipFree(c->d_status);
    (void)hipFree(c->d_conns_py);
    (void)hipFree(c->d_persons);
    (void)hipFree(c->d_npersons);
    (void)hipFree(c->d_records);
    (void)hipFree(c->d_paf);
    delete c;
}

int check_shape(const pp_ctx *c, int batch, int dtype, int h, int w) {
    if (!c || batch <= 0 || h <= 0 || w <= 0 || (dtype != PP_F16 && dtype != PP_F32)) return PP_ERR_BAD_ARG;
    if (batch > c->max_batch || (size_t)h * w > (size_t)c->max_h * c->max_w) return PP_ERR_TOO_LARGE;
    const int elem = dtype == PP_F16 ? 2 : 4;
    if (pp::lds_bytes_heat(elem, h, w, c->maxp) > pp::kMaxDynLds) return PP_ERR_TOO_LARGE;
    if (pp::lds_bytes_limb(elem, h, w, c->maxp, c->cap) > pp::kMaxDynLds) return PP_ERR_TOO_LARGE;
    return PP_OK;
}

}  // namespace

extern "C" {

int pp_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

const char *pp_status_string(int status) {
    switch (status) {
        case PP_OK: return "ok";
        case PP_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
        case PP_ERR_BAD_ARG: return "bad argument";
        case PP_ERR_TOO_LARGE: return "batch or map size beyond the context's capacity / LDS";
        case PP_ERR_HIP: return "HIP runtime error (see pp_last_hip_error)";
        case PP_ERR_OVERFLOW: return "capacity exceeded (peaks per part or humans)";
        case PP_ERR_UNSUPPORTED: return "shape not supported by this convolution tile configuration";
        default: return "unknown status";
    }
}

int pp_create(pp_ctx **out, int device, int max_batch, int max_h, int max_w, int max_peaks_per_part) {
    if (!out || max_batch <= 0 || max_h <= 0 || max_w <= 0 || max_peaks_per_part <= 0 ||
        max_peaks_per_part > PP_MAX_PEAKS_PER_PART_LIMIT)
        return PP_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PP_ERR_NO_DEVICE;
    pp_ctx *c = new pp_ctx();
    c->device = device;
    c->max_batch = max_batch;
    c->max_h = max_h;
    c->max_w = max_w;
    c->maxp = max_peaks_per_part;
    // candidate capacity per (limb, image); never below one workgroup round (256 pairs): K_B parks a round's survivors in
    // the candidate-sized scratch arrays
    c->cap = max_peaks_per_part * max_peaks_per_part < 512 ? max_peaks_per_part * max_peaks_per_part : 512;
    if (c->cap < 256) c->cap = 256;
    hipError_t e = pp::init_kernel_attributes();
    const size_t B = (size_t)max_batch;
    if (e == hipSuccess) e = hipMalloc(&c->d_peaks, B * PP_NUM_PART * c->maxp * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&c->d_counts, B * PP_NUM_PART * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&c->d_conns, B * PP_NUM_LIMB * c->maxp * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&c->d_conn_counts, B * PP_NUM_LIMB * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&c->d_conn_aux, B * PP_NUM_LIMB * c->maxp * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&c->d_sync, (2 * B + 16 + PP_NUM_LIMB * B) * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_sync, 0, (2 * B + 16 + PP_NUM_LIMB * B) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&c->d_conns_py, B * PP_NUM_LIMB * c->maxp * 32);
    if (e == hipSuccess) e = hipMalloc(&c->d_persons, 128 * 40 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&c->d_npersons, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&c->d_status, B * pp::kFlagWordsPerImage * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&c->d_records, B * sizeof(pp_record));
    if (e == hipSuccess) e = hipMemset(c->d_counts, 0, B * PP_NUM_PART * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_conn_counts, 0, B * PP_NUM_LIMB * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_status, 0, B * pp::kFlagWordsPerImage * sizeof(unsigned));
    if (e != hipSuccess) {
        free_ctx(c);
        return PP_ERR_HIP;
    }
    *out = c;
    return PP_OK;
}

int pp_destroy(pp_ctx *ctx) {
    if (!ctx) return PP_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    free_ctx(ctx);
    return PP_OK;
}

int pp_last_hip_error(const pp_ctx *ctx) { return ctx ? ctx->hip_err : 0; }

int pp_debug_set_mode(pp_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return PP_ERR_BAD_ARG;
    ctx->mode = mode;
    return PP_OK;
}

int pp_debug_set_stamps(long long *stamps_dev) {
    return pp::set_stamp_buffer(stamps_dev) == hipSuccess ? PP_OK : PP_ERR_HIP;
}

int pp_nms_batch(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip, int refine,
                 float *peaks_dev, int *counts_dev, void *stream) {
    return pp_nms_batch_ex(ctx, batch, net_out_dev, dtype, h, w, flip, 0, 0.1f, refine ? 1 : 0, peaks_dev, counts_dev, stream);
}

int pp_nms_batch_ex(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip, int nms_mode,
                    float threshold, int refine_mode, float *peaks_dev, int *counts_dev, void *stream) {
    int rc = check_shape(ctx, batch, dtype, h, w);
    if (rc != PP_OK) return rc;
    if (!net_out_dev || nms_mode < 0 || nms_mode > 1 || refine_mode < 0 || refine_mode > 3) return PP_ERR_BAD_ARG;
    const int refine = refine_mode;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float4 *pk = peaks_dev ? reinterpret_cast<float4 *>(peaks_dev) : ctx->d_peaks;
    int *cn = counts_dev ? counts_dev : ctx->d_counts;
    PP_HIP(ctx, pp::launch_heat_peaks(net_out_dev, dtype, batch, flip ? 2 : 1, h, w, flip, refine, nms_mode, threshold,
                                      ctx->maxp, pk, cn, ctx->d_status, nullptr, nullptr, st));
    ctx->last_stream = st;
    ctx->last_batch = batch;
    ctx->last_peaks = pk;
    ctx->last_counts = cn;
    return PP_OK;
}

int pp_process_batch(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip,
                     int min_img_size, const int *min_img_size_dev, pp_record *records_dev, void *stream) {
    int rc = check_shape(ctx, batch, dtype, h, w);
    if (rc != PP_OK) return rc;
    if (!net_out_dev) return PP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    pp_record *rec = records_dev ? records_dev : ctx->d_records;
    const int ns = flip ? 2 : 1;
    // two launches: K_A (peaks; its last workgroup orders the images by load), K_B (limb scoring + matching; the last limb
    // workgroup of each image assembles it).  ctx->mode: 0 as described, 1 = K_B and the assembly as separate launches,
    // 2 = fused without the load ordering (A/B measurements, pp_debug_set_mode).
    int *order = ctx->d_sync, *arrive = ctx->d_sync + ctx->max_batch, *arrive_all = ctx->d_sync + 2 * ctx->max_batch;
    const bool sorted = ctx->mode == 0 && pp::heat_peaks_sorts(dtype, batch, h, w, ctx->maxp);
    PP_HIP(ctx, pp::launch_heat_peaks(net_out_dev, dtype, batch, ns, h, w, flip, 1, 0, 0.1f, ctx->maxp, ctx->d_peaks,
                                      ctx->d_counts, ctx->d_status, sorted ? order : nullptr, sorted ? arrive_all : nullptr, st));
    PP_HIP(ctx, pp::launch_limb_connect(net_out_dev, dtype, batch, ns, h, w, flip, ctx->maxp, ctx->cap, min_img_size,
                                        min_img_size_dev, ctx->d_peaks, ctx->d_counts, ctx->d_conns, ctx->d_conn_aux,
                                        ctx->d_conn_counts, ctx->d_status, sorted ? order : nullptr,
                                        ctx->mode == 1 ? nullptr : arrive,
                                        reinterpret_cast<unsigned *>(ctx->d_sync + 2 * ctx->max_batch + 16), rec, st));
    if (ctx->mode == 1)
        PP_HIP(ctx, pp::launch_assemble_wave(batch, ctx->maxp, ctx->d_peaks, ctx->d_counts, ctx->d_conns, ctx->d_conn_aux,
                                             ctx->d_conn_counts, ctx->d_status, rec, st));
    ctx->last_stream = st;
    ctx->last_batch = batch;
    ctx->last_peaks = ctx->d_peaks;
    ctx->last_counts = ctx->d_counts;
    return PP_OK;
}

int pp_process_batch_py(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip, int img_height,
                        const int *img_height_dev, pp_record *records_dev, void *stream) {
    int rc = check_shape(ctx, batch, dtype, h, w);
    if (rc != PP_OK) return rc;
    if (!net_out_dev) return PP_ERR_BAD_ARG;
    const int elem = dtype == PP_F16 ? 2 : 4;
    if (pp::lds_bytes_limb_py(elem, h, w, ctx->maxp, ctx->cap) > pp::kMaxDynLds ||
        pp::lds_bytes_assemble_py(ctx->maxp) > pp::kMaxDynLds)
        return PP_ERR_TOO_LARGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    pp_record *rec = records_dev ? records_dev : ctx->d_records;
    const int ns = flip ? 2 : 1;
    PP_HIP(ctx, pp::launch_heat_peaks(net_out_dev, dtype, batch, ns, h, w, flip, 1, 0, 0.1f, ctx->maxp, ctx->d_peaks,
                                      ctx->d_counts, ctx->d_status, nullptr, nullptr, st));
    PP_HIP(ctx, pp::launch_limb_connect_py(net_out_dev, dtype, batch, ns, h, w, flip, ctx->maxp, ctx->cap, img_height,
                                           img_height_dev, ctx->d_peaks, ctx->d_counts, ctx->d_conns_py, ctx->d_conn_counts,
                                           ctx->d_status, st));
    PP_HIP(ctx, pp::launch_assemble_py(batch, ctx->maxp, 0, ctx->d_peaks, ctx->d_counts, ctx->d_conns_py, ctx->d_conn_counts,
                                       ctx->d_status, 0, rec, nullptr, nullptr, st));
    ctx->last_stream = st;
    ctx->last_batch = batch;
    ctx->last_peaks = ctx->d_peaks;
    ctx->last_counts = ctx->d_counts;
    return PP_OK;
}

int pp_time_kernels(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip,
                    int min_img_size, int iters, float *ms_out, void *stream) {
    int rc = check_shape(ctx, batch, dtype, h, w);
    if (rc != PP_OK) return rc;
    if (!net_out_dev || !ms_out || iters <= 0) return PP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ns = flip ? 2 : 1;
    int *order = ctx->d_sync, *arrive = ctx->d_sync + ctx->max_batch, *arrive_all = ctx->d_sync + 2 * ctx->max_batch;
    const bool sorted = ctx->mode == 0 && pp::heat_peaks_sorts(dtype, batch, h, w, ctx->maxp);
    hipEvent_t e0, e1;
    PP_HIP(ctx, hipEventCreate(&e0));
    PP_HIP(ctx, hipEventCreate(&e1));
    // one untimed pass so that every kernel's inputs exist
    rc = pp_process_batch(ctx, batch, net_out_dev, dtype, h, w, flip, min_img_size, nullptr, nullptr, st);
    if (rc != PP_OK) return rc;
    for (int k = 0; k < 4; k++) {
        PP_HIP(ctx, hipEventRecord(e0, st));
        for (int i = 0; i < iters; i++) {
            if (k == 3) {  // the whole chain, back to back, as pp_process_batch enqueues it
                rc = pp_process_batch(ctx, batch, net_out_dev, dtype, h, w, flip, min_img_size, nullptr, nullptr, st);
                if (rc != PP_OK) return rc;
            } else if (k == 0) {
                PP_HIP(ctx, pp::launch_heat_peaks(net_out_dev, dtype, batch, ns, h, w, flip, 1, 0, 0.1f, ctx->maxp,
                                                  ctx->d_peaks, ctx->d_counts, ctx->d_status, sorted ? order : nullptr,
                                                  sorted ? arrive_all : nullptr, st));
            } else if (k == 1) {  // limb scoring + matching with the assembly tail (the product's second launch)
                PP_HIP(ctx, pp::launch_limb_connect(net_out_dev, dtype, batch, ns, h, w, flip, ctx->maxp, ctx->cap,
                                                    min_img_size, nullptr, ctx->d_peaks, ctx->d_counts, ctx->d_conns,
                                                    ctx->d_conn_aux, ctx->d_conn_counts, ctx->d_status,
                                                    sorted ? order : nullptr, arrive,
                                                    reinterpret_cast<unsigned *>(ctx->d_sync + 2 * ctx->max_batch + 16), ctx->d_records, st));
            } else {  // the assembly alone, one wave per image, as its own launch (diagnostic)
                PP_HIP(ctx, pp::launch_assemble_wave(batch, ctx->maxp, ctx->d_peaks, ctx->d_counts, ctx->d_conns,
                                                     ctx->d_conn_aux, ctx->d_conn_counts, ctx->d_status, ctx->d_records, st));
            }
        }
        PP_HIP(ctx, hipEventRecord(e1, st));
        PP_HIP(ctx, hipEventSynchronize(e1));
        float ms = 0.f;
        PP_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        ms_out[k] = ms / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PP_OK;
}

int pp_read_peaks(pp_ctx *ctx, int image, float *joint_list_host, int max_rows, int *n_rows) {
    if (!ctx || !n_rows || image < 0 || image >= ctx->last_batch || !ctx->last_peaks) return PP_ERR_BAD_ARG;
    PP_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    int counts[PP_NUM_PART];
    PP_HIP(ctx, hipMemcpy(counts, ctx->last_counts + (size_t)image * PP_NUM_PART, sizeof(counts), hipMemcpyDeviceToHost));
    std::vector<float4> pk((size_t)PP_NUM_PART * ctx->maxp);
    PP_HIP(ctx, hipMemcpy(pk.data(), ctx->last_peaks + (size_t)image * PP_NUM_PART * ctx->maxp, pk.size() * sizeof(float4),
                          hipMemcpyDeviceToHost));
    int n = 0;
    bool truncated = false;
    for (int part = 0; part < PP_NUM_PART; part++) {
        if (counts[part] > ctx->maxp) truncated = true;
        const int c = counts[part] < ctx->maxp ? counts[part] : ctx->maxp;
        for (int r = 0; r < c; r++, n++) {
            if (joint_list_host && n < max_rows) {
                const float4 p = pk[(size_t)part * ctx->maxp + r];
                float *row = joint_list_host + (size_t)5 * n;
                row[0] = p.x;
                row[1] = p.y;
                row[2] = p.z;
                row[3] = (float)n;     // cnt_total_joints, utils/parse_skeletons.py:172-173
                row[4] = (float)part;  // evaluate.py:99-103
            }
        }
    }
    *n_rows = n;
    return truncated ? PP_ERR_OVERFLOW : PP_OK;  // rows are valid but a part had more peaks than the context holds
}

int pp_read_connections(pp_ctx *ctx, int image, int limb, float *rows_host, int max_rows, int *n_rows) {
    if (!ctx || !n_rows || image < 0 || image >= ctx->last_batch || limb < 0 || limb >= PP_NUM_LIMB) return PP_ERR_BAD_ARG;
    PP_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    int n = 0;
    PP_HIP(ctx, hipMemcpy(&n, ctx->d_conn_counts + (size_t)image * PP_NUM_LIMB + limb, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<float4> cn((size_t)ctx->maxp);
    PP_HIP(ctx, hipMemcpy(cn.data(), ctx->d_conns + ((size_t)image * PP_NUM_LIMB + limb) * ctx->maxp,
                          cn.size() * sizeof(float4), hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < max_rows && rows_host; i++) {
        int a, b;
        std::memcpy(&a, &cn[i].x, 4);
        std::memcpy(&b, &cn[i].y, 4);
        rows_host[4 * i + 0] = (float)a;
        rows_host[4 * i + 1] = (float)b;
        rows_host[4 * i + 2] = cn[i].z;
        rows_host[4 * i + 3] = cn[i].w;
    }
    *n_rows = n;
    return PP_OK;
}

int pp_read_part_counts(pp_ctx *ctx, int image, int *counts_host) {
    if (!ctx || !counts_host || image < 0 || image >= ctx->last_batch) return PP_ERR_BAD_ARG;
    PP_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    PP_HIP(ctx, hipMemcpy(counts_host, ctx->d_counts + (size_t)image * PP_NUM_PART, sizeof(int) * PP_NUM_PART,
                          hipMemcpyDeviceToHost));
    return PP_OK;
}

int pp_read_connection_counts(pp_ctx *ctx, int image, int *counts_host) {
    if (!ctx || !counts_host || image < 0 || image >= ctx->last_batch) return PP_ERR_BAD_ARG;
    PP_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    PP_HIP(ctx, hipMemcpy(counts_host, ctx->d_conn_counts + (size_t)image * PP_NUM_LIMB, sizeof(int) * PP_NUM_LIMB,
                          hipMemcpyDeviceToHost));
    return PP_OK;
}

int pp_debug_read_flags(pp_ctx *ctx, int image, uint32_t *flags_host) {
    if (!ctx || !flags_host || image < 0 || image >= ctx->max_batch) return PP_ERR_BAD_ARG;
    PP_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    PP_HIP(ctx, hipMemcpy(flags_host, ctx->d_status + (size_t)image * pp::kFlagWordsPerImage,
                          sizeof(uint32_t) * pp::kFlagWordsPerImage, hipMemcpyDeviceToHost));
    return PP_OK;
}

int pp_read_records(pp_ctx *ctx, const pp_record *records_dev, pp_record *records_host, int batch, void *stream) {
    if (!ctx || !records_host || batch <= 0 || batch > ctx->max_batch) return PP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const pp_record *src = records_dev ? records_dev : ctx->d_records;
    PP_HIP(ctx, hipMemcpyAsync(records_host, src, sizeof(pp_record) * (size_t)batch, hipMemcpyDeviceToHost, st));
    PP_HIP(ctx, hipStreamSynchronize(st));
    return PP_OK;
}

// ------------------------------------------------------------------------------------------------ drop-in
int pp_process_paf_host(pp_ctx *ctx, int p1, int p2, int p3, const float *peaks, int f1, int f2, int f3,
                        const float *pafmap, int min_img_size) {
    if (!ctx || !peaks || !pafmap || p1 < 0 || p2 < 0 || p3 < 5 || f1 <= 0 || f2 <= 0 || f3 <= 0) return PP_ERR_BAD_ARG;
    ctx->have_result = false;
    const int maxp = ctx->maxp;
    // pafprocess.cpp:29-41: bucket rows by part, ids in input order, x/y truncated to int
    std::vector<float4> pk((size_t)PP_NUM_PART * maxp);
    int counts[PP_NUM_PART] = {0};
    int peak_cnt = 0;
    for (int i = 0; i < p1; i++) {
        for (int j = 0; j < p2; j++) {
            const float *r = peaks + (size_t)p3 * (j + (size_t)p2 * i);
            const int id = peak_cnt++;
            const int part = (int)r[4];
            if (part < 0 || part >= PP_NUM_PART) return PP_ERR_BAD_ARG;  // the reference writes out of bounds here
            if (counts[part] >= maxp) return PP_ERR_OVERFLOW;
            float4 v;
            v.x = (float)(int)r[0];
            v.y = (float)(int)r[1];
            v.z = r[2];
            std::memcpy(&v.w, &id, 4);
            pk[(size_t)part * maxp + counts[part]++] = v;
        }
    }
    // pafprocess.cpp:43-48: flattened table for the getters
    ctx->line_x.clear();
    ctx->line_y.clear();
    ctx->line_s.clear();
    for (int part = 0; part < PP_NUM_PART; part++)
        for (int r = 0; r < counts[part]; r++) {
            const float4 v = pk[(size_t)part * maxp + r];
            ctx->line_x.push_back((int)v.x);
            ctx->line_y.push_back((int)v.y);
            ctx->line_s.push_back(v.z);
        }
    PP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t paf_bytes = sizeof(float) * (size_t)f1 * f2 * f3;
    if (paf_bytes > ctx->d_paf_bytes) {
        if (ctx->d_paf) (void)hipFree(ctx->d_paf);
        ctx->d_paf = nullptr;
        ctx->d_paf_bytes = 0;
        PP_HIP(ctx, hipMalloc(&ctx->d_paf, paf_bytes));
        ctx->d_paf_bytes = paf_bytes;
    }
    hipStream_t st = nullptr;
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_paf, pafmap, paf_bytes, hipMemcpyHostToDevice, st));
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_peaks, pk.data(), pk.size() * sizeof(float4), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_counts, counts, sizeof(counts), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, pp::launch_limb_connect_hwc(ctx->d_paf, f1, f2, f3, maxp, ctx->cap, min_img_size, ctx->d_peaks,
                                            ctx->d_counts, ctx->d_conns, ctx->d_conn_counts, ctx->d_status, st));
    PP_HIP(ctx, pp::launch_assemble(1, maxp, 1, ctx->d_peaks, ctx->d_counts, ctx->d_conns, ctx->d_conn_counts,
                                    ctx->d_status, PP_NUM_PART, ctx->d_records, st));  // no peak kernel ran: limb words only
    PP_HIP(ctx, hipMemcpyAsync(&ctx->h_record, ctx->d_records, sizeof(pp_record), hipMemcpyDeviceToHost, st));
    PP_HIP(ctx, hipStreamSynchronize(st));
    ctx->last_stream = st;
    ctx->last_batch = 1;
    ctx->last_peaks = ctx->d_peaks;
    ctx->last_counts = ctx->d_counts;
    if (ctx->h_record.status & (PP_ST_HUMAN_OVERFLOW | PP_ST_SKEL_OVERFLOW | PP_ST_CAND_OVERFLOW)) return PP_ERR_OVERFLOW;
    ctx->have_result = true;
    return PP_OK;
}

int pp_get_num_humans(const pp_ctx *ctx) { return (ctx && ctx->have_result) ? ctx->h_record.n_humans : 0; }
int pp_get_part_peak_id(const pp_ctx *ctx, int skeleton_id, int part_id) {
    if (!ctx || !ctx->have_result || skeleton_id < 0 || skeleton_id >= ctx->h_record.n_humans || part_id < 0 ||
        part_id >= PP_NUM_PART)
        return -1;
    return ctx->h_record.humans[skeleton_id].peak_id[part_id];
}
float pp_get_score(const pp_ctx *ctx, int skeleton_id) {
    if (!ctx || !ctx->have_result || skeleton_id < 0 || skeleton_id >= ctx->h_record.n_humans) return 0.0f;
    return ctx->h_record.humans[skeleton_id].score;
}
int pp_get_part_x(const pp_ctx *ctx, int cid) {
    return (ctx && cid >= 0 && cid < (int)ctx->line_x.size()) ? ctx->line_x[cid] : 0;
}
int pp_get_part_y(const pp_ctx *ctx, int cid) {
    return (ctx && cid >= 0 && cid < (int)ctx->line_y.size()) ? ctx->line_y[cid] : 0;
}
float pp_get_part_score(const pp_ctx *ctx, int cid) {
    return (ctx && cid >= 0 && cid < (int)ctx->line_s.size()) ? ctx->line_s[cid] : 0.0f;
}
uint32_t pp_get_status(const pp_ctx *ctx) { return (ctx && ctx->have_result) ? ctx->h_record.status : 0u; }

// ------------------------------------------------------------------------------------------------ original path (A10)
int pp_original_accumulate(pp_ctx *ctx, int batch, const void *net_out_dev, int dtype, int h, int w, int flip, int pad_down,
                           int pad_right, int img_h, int img_w, int n_scales, float *scratch_planar, float *scratch_up,
                           double *heat_acc, double *paf_acc, void *stream) {
    if (!ctx || !net_out_dev || !scratch_planar || !scratch_up || !heat_acc || !paf_acc || batch <= 0 || h <= 0 || w <= 0 ||
        img_h <= 0 || img_w <= 0 || n_scales <= 0 || pad_down < 0 || pad_right < 0 || pad_down >= 4 * h || pad_right >= 4 * w ||
        (dtype != PP_F16 && dtype != PP_F32))
        return PP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int uh = 4 * h, uw = 4 * w, ch = uh - pad_down, cw = uw - pad_right;
    PP_HIP(ctx, pp::launch_flip_average_planar(net_out_dev, dtype, batch, h, w, flip, scratch_planar, st));
    // x4 up-sample of every (image, channel) plane (utils/parse_skeletons.py:252-263)
    PP_HIP(ctx, pp::launch_resize_cubic(scratch_planar, (long)h * w, w, h, w, scratch_up, 0, batch * PP_NUM_CH, uh, uw, 0.25, 0.25,
                                        1.0f, st));
    // crop the padding (:272-273), resize to the image size (:276-277), accumulate value / n_scales in float64 (:280-281)
    const double sx = 1.0 / ((double)img_w / (double)cw), sy = 1.0 / ((double)img_h / (double)ch);
    for (int b = 0; b < batch; b++) {
        const float *up_b = scratch_up + (size_t)b * PP_NUM_CH * uh * uw;
        PP_HIP(ctx, pp::launch_resize_cubic(up_b, (long)uh * uw, uw, ch, cw, paf_acc + (size_t)b * PP_NUM_LIMB * img_h * img_w, 1,
                                            PP_NUM_LIMB, img_h, img_w, sx, sy, (float)n_scales, st));
        PP_HIP(ctx, pp::launch_resize_cubic(up_b + (size_t)PP_NUM_LIMB * uh * uw, (long)uh * uw, uw, ch, cw,
                                            heat_acc + (size_t)b * PP_NUM_HEAT * img_h * img_w, 1, PP_NUM_HEAT, img_h, img_w, sx,
                                            sy, (float)n_scales, st));
    }
    return PP_OK;
}

int pp_original_accumulate_all(pp_ctx *ctx, int batch, int n_scales, const void *const *net_out_dev, int dtype, const int *h,
                               const int *w, int flip, const int *pad_down, const int *pad_right, int img_h, int img_w,
                               double *heat_acc, double *paf_acc, void *stream) {
    if (!ctx || !net_out_dev || !h || !w || !pad_down || !pad_right || !heat_acc || !paf_acc || batch <= 0 || img_h <= 0 ||
        img_w <= 0 || n_scales <= 0 || (dtype != PP_F16 && dtype != PP_F32))
        return PP_ERR_BAD_ARG;
    if (n_scales > 6) return PP_ERR_UNSUPPORTED;
    for (int i = 0; i < n_scales; i++)
        if (!net_out_dev[i] || h[i] <= 0 || w[i] <= 0 || pad_down[i] < 0 || pad_right[i] < 0 || pad_down[i] >= 4 * h[i] ||
            pad_right[i] >= 4 * w[i])
            return PP_ERR_BAD_ARG;
    const hipError_t e = pp::launch_accumulate_scales(n_scales, net_out_dev, dtype, batch, h, w, flip, pad_down, pad_right, img_h,
                                                      img_w, heat_acc, paf_acc, static_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return PP_ERR_UNSUPPORTED;   // a scale whose tiles do not fit LDS: use pp_original_accumulate
    PP_HIP(ctx, e);
    return PP_OK;
}

int pp_original_finish(pp_ctx *ctx, int batch, int img_h, int img_w, float thre1, const double *heat_acc, const double *paf_acc,
                       unsigned char *mask_scratch, void *peaks64_scratch, pp_record *records_dev, void *stream) {
    if (!ctx || !heat_acc || !paf_acc || !mask_scratch || !peaks64_scratch || batch <= 0 || batch > ctx->max_batch ||
        img_h <= 0 || img_w <= 0)
        return PP_ERR_BAD_ARG;
    if (pp::lds_bytes_assemble_py(ctx->maxp) > pp::kMaxDynLds) return PP_ERR_TOO_LARGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    pp_record *rec = records_dev ? records_dev : ctx->d_records;
    PP_HIP(ctx, pp::launch_fullres(batch, img_h, img_w, thre1, ctx->maxp, ctx->cap, img_h, heat_acc, paf_acc, mask_scratch,
                                   peaks64_scratch, ctx->d_counts, ctx->d_conns_py, ctx->d_conn_counts, ctx->d_status, rec, st));
    ctx->last_stream = st;
    ctx->last_batch = batch;
    return PP_OK;
}

// ------------------------------------------------------------------------------------------------ Python twins, host form
namespace {
// all_peaks[k] in input order (what find_connections indexes with i / j); rows keep their own id (column 3)
int bucket_py(const pp_ctx *ctx, const float *peaks, int n, std::vector<float4> &pk, int counts[PP_NUM_PART]) {
    const int maxp = ctx->maxp;
    pk.assign((size_t)PP_NUM_PART * maxp, float4{0, 0, 0, 0});
    for (int k = 0; k < PP_NUM_PART; k++) counts[k] = 0;
    for (int i = 0; i < n; i++) {
        const float *r = peaks + (size_t)5 * i;
        const int part = (int)r[4];
        if (part < 0 || part >= PP_NUM_PART) return PP_ERR_BAD_ARG;
        if (counts[part] >= maxp) return PP_ERR_OVERFLOW;
        float4 v;
        v.x = r[0];
        v.y = r[1];
        v.z = r[2];
        v.w = r[3];  // id as given (a float in the joint list)
        pk[(size_t)part * maxp + counts[part]++] = v;
    }
    return PP_OK;
}
}  // namespace

int pp_py_find_connections_host(pp_ctx *ctx, const float *peaks, int n, const float *paf, int H, int W, int C, int img_height,
                                double *conns_out, int *counts_out, int *special_out) {
    if (!ctx || !peaks || !paf || !conns_out || !counts_out || n < 0 || H <= 0 || W <= 0 || C <= 0) return PP_ERR_BAD_ARG;
    if (pp::lds_bytes_assemble_py(ctx->maxp) > pp::kMaxDynLds) return PP_ERR_TOO_LARGE;
    std::vector<float4> pk;
    int counts[PP_NUM_PART];
    int rc = bucket_py(ctx, peaks, n, pk, counts);
    if (rc != PP_OK) return rc;
    const int maxp = ctx->maxp;
    PP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t paf_bytes = sizeof(float) * (size_t)H * W * C;
    if (paf_bytes > ctx->d_paf_bytes) {
        if (ctx->d_paf) (void)hipFree(ctx->d_paf);
        ctx->d_paf = nullptr;
        ctx->d_paf_bytes = 0;
        PP_HIP(ctx, hipMalloc(&ctx->d_paf, paf_bytes));
        ctx->d_paf_bytes = paf_bytes;
    }
    hipStream_t st = nullptr;
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_paf, paf, paf_bytes, hipMemcpyHostToDevice, st));
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_peaks, pk.data(), pk.size() * sizeof(float4), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_counts, counts, sizeof(counts), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, pp::launch_limb_connect_py_hwc(ctx->d_paf, H, W, C, maxp, ctx->cap, img_height, ctx->d_peaks, ctx->d_counts,
                                               ctx->d_conns_py, ctx->d_conn_counts, ctx->d_status, st));
    std::vector<double> raw((size_t)PP_NUM_LIMB * maxp * 4);
    PP_HIP(ctx, hipMemcpyAsync(raw.data(), ctx->d_conns_py, raw.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    PP_HIP(ctx, hipMemcpyAsync(counts_out, ctx->d_conn_counts, sizeof(int) * PP_NUM_LIMB, hipMemcpyDeviceToHost, st));
    unsigned limb_flags[PP_NUM_LIMB];
    PP_HIP(ctx, hipMemcpyAsync(limb_flags, ctx->d_status + PP_NUM_PART, sizeof(limb_flags), hipMemcpyDeviceToHost, st));
    PP_HIP(ctx, hipStreamSynchronize(st));
    unsigned status = 0;
    for (unsigned f : limb_flags) status |= f;
    if (status & PP_ST_CAND_OVERFLOW) return PP_ERR_OVERFLOW;
    static const int LP[PP_NUM_LIMB][2] = {{1, 0},   {1, 14},  {1, 15},  {1, 16},  {1, 17},  {0, 14},  {0, 15},  {14, 16},
                                           {15, 17}, {1, 2},   {2, 3},   {3, 4},   {1, 5},   {5, 6},   {6, 7},   {1, 8},
                                           {8, 9},   {9, 10},  {1, 11},  {11, 12}, {12, 13}, {0, 2},   {0, 5},   {2, 8},
                                           {8, 12},  {5, 11},  {11, 9},  {16, 2},  {17, 5},  {8, 11}};
    for (int limb = 0; limb < PP_NUM_LIMB; limb++) {
        const int pa = LP[limb][0], pb = LP[limb][1];
        if (special_out) special_out[limb] = (counts[pa] == 0 && counts[pb] == 0) ? 1 : 0;  // parse_skeletons.py:340-342
        for (int k = 0; k < counts_out[limb]; k++) {
            const double *r = raw.data() + ((size_t)limb * maxp + k) * 4;
            double *o = conns_out + ((size_t)limb * maxp + k) * 6;
            const int i = (int)r[0], j = (int)r[1];
            o[0] = (double)pk[(size_t)pa * maxp + i].w;  // joints_src[i][3]
            o[1] = (double)pk[(size_t)pb * maxp + j].w;
            o[2] = r[2];
            o[3] = (double)i;
            o[4] = (double)j;
            o[5] = r[3];
        }
    }
    return PP_OK;
}

int pp_py_find_humans_host(pp_ctx *ctx, const double *conns, const int *counts_limb, const float *peaks, int n,
                           double *persons_out, int cap, int *n_out) {
    if (!ctx || !conns || !counts_limb || !peaks || !persons_out || !n_out || n < 0 || cap <= 0) return PP_ERR_BAD_ARG;
    if (pp::lds_bytes_assemble_py(ctx->maxp) > pp::kMaxDynLds) return PP_ERR_TOO_LARGE;
    std::vector<float4> pk;
    int counts[PP_NUM_PART];
    int rc = bucket_py(ctx, peaks, n, pk, counts);
    if (rc != PP_OK) return rc;
    const int maxp = ctx->maxp;
    std::vector<double> raw((size_t)PP_NUM_LIMB * maxp * 4, 0.0);
    int cl[PP_NUM_LIMB];
    for (int limb = 0; limb < PP_NUM_LIMB; limb++) {
        cl[limb] = counts_limb[limb];
        if (cl[limb] < 0 || cl[limb] > maxp) return PP_ERR_BAD_ARG;
        for (int k = 0; k < cl[limb]; k++) {
            const double *i6 = conns + ((size_t)limb * maxp + k) * 6;
            double *r = raw.data() + ((size_t)limb * maxp + k) * 4;
            r[0] = i6[0];  // src peak id
            r[1] = i6[1];  // dst peak id
            r[2] = i6[2];
            r[3] = i6[5];
        }
    }
    PP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_peaks, pk.data(), pk.size() * sizeof(float4), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_counts, counts, sizeof(counts), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_conns_py, raw.data(), raw.size() * sizeof(double), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, hipMemcpyAsync(ctx->d_conn_counts, cl, sizeof(cl), hipMemcpyHostToDevice, st));
    PP_HIP(ctx, pp::launch_assemble_py(1, maxp, 1, ctx->d_peaks, ctx->d_counts, ctx->d_conns_py, ctx->d_conn_counts,
                                       ctx->d_status, pp::kFlagWordsPerImage, ctx->d_records, ctx->d_persons, ctx->d_npersons,
                                       st));  // connections come from the caller: no kernel wrote a flag word
    int np = 0;
    PP_HIP(ctx, hipMemcpyAsync(&np, ctx->d_npersons, sizeof(int), hipMemcpyDeviceToHost, st));
    PP_HIP(ctx, hipStreamSynchronize(st));
    if (np > 128 || np > cap) return PP_ERR_OVERFLOW;
    PP_HIP(ctx, hipMemcpy(persons_out, ctx->d_persons, sizeof(double) * 40 * (size_t)np, hipMemcpyDeviceToHost));
    *n_out = np;
    return PP_OK;
}

// ---- the reference's seven names, one process-wide context (pafprocess.cpp:16-17 keeps globals too)
static pp_ctx *g_ctx = nullptr;
static std::mutex g_mu;

int process_paf(int p1, int p2, int p3, float *peaks, int f1, int f2, int f3, float *pafmap, int min_img_size) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx) {
        int dev = 0;
        if (const char *e = std::getenv("POSEPAF_DEVICE")) dev = std::atoi(e);
        int rc = pp_create(&g_ctx, dev, 1, 128, 128, PP_MAX_PEAKS_PER_PART_LIMIT);
        if (rc != PP_OK) {
            std::fprintf(stderr, "posepaf: process_paf cannot run: %s\n", pp_status_string(rc));
            return rc;
        }
    }
    int rc = pp_process_paf_host(g_ctx, p1, p2, p3, peaks, f1, f2, f3, pafmap, min_img_size);
    if (rc != PP_OK) std::fprintf(stderr, "posepaf: process_paf failed: %s\n", pp_status_string(rc));
    return rc;
}
int get_num_humans(void) { return pp_get_num_humans(g_ctx); }
int get_part_peak_id(int skeleton_id, int part_id) { return pp_get_part_peak_id(g_ctx, skeleton_id, part_id); }
float get_score(int skeleton_id) { return pp_get_score(g_ctx, skeleton_id); }
int get_part_x(int cid) { return pp_get_part_x(g_ctx, cid); }
int get_part_y(int cid) { return pp_get_part_y(g_ctx, cid); }
float get_part_score(int cid) { return pp_get_part_score(g_ctx, cid); }

}  // extern "C"
