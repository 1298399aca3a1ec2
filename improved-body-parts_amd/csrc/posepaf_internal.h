// posepaf_internal.h -- declarations shared by the kernels and the C-ABI host code (not installed).
#ifndef POSEPAF_INTERNAL_H
#define POSEPAF_INTERNAL_H

#include <hip/hip_runtime.h>

#include "../../include/posepaf.h"

namespace pp {

// 160 KiB of LDS per CU (gfx950) minus room for the kernels' static __shared__ arrays (< 2 KB)
constexpr size_t kMaxDynLds = 163840 - 4096;
hipError_t init_kernel_attributes();
hipError_t set_stamp_buffer(long long *buf);

// status flag words per image: [0, 18) one per part (peak kernels), [18, 48) one per limb (limb kernels); see or_flags
constexpr int kFlagWordsPerImage = PP_NUM_PART + PP_NUM_LIMB;

size_t lds_bytes_heat(int elem, int h, int w, int maxp);
size_t lds_bytes_limb(int elem, int h, int w, int maxp, int cap);
size_t lds_bytes_limb_hwc(int maxp, int cap);
size_t lds_bytes_assemble(int maxp);

// order / arrive_all (both or neither): K_A's last workgroup writes the images sorted by matching load into order[batch]
hipError_t launch_heat_peaks(const void *net, int dtype, int batch, int n_samples, int h, int w, int flip, int refine,
                             int nms_mode, float thr, int maxp, float4 *peaks, int *counts, unsigned *status, int *order,
                             int *arrive_all, hipStream_t stream);
bool heat_peaks_sorts(int dtype, int batch, int h, int w, int maxp);
// arrive != NULL: fused form, workgroup 30 of each image assembles it into records WHILE its limbs are matched (arrive[batch]:
// per-image launch counters, ready[batch][30]: per-limb publication flags; both zeroed once at create and owned by the kernel);
// arrive == NULL: connections only (launch_assemble_wave follows).  order may be NULL.
hipError_t launch_limb_connect(const void *net, int dtype, int batch, int n_samples, int h, int w, int flip, int maxp,
                               int cap, int min_img_size, const int *min_img_size_dev, const float4 *peaks,
                               const int *counts, float4 *conns, float4 *aux, int *conn_counts, unsigned *status,
                               const int *order, int *arrive, unsigned *ready, pp_record *records, hipStream_t stream);
hipError_t launch_assemble_wave(int batch, int maxp, const float4 *peaks, const int *counts, const float4 *conns,
                                const float4 *aux, const int *conn_counts, const unsigned *status, pp_record *records,
                                hipStream_t stream);
hipError_t launch_limb_connect_hwc(const float *paf, int H, int W, int C, int maxp, int cap, int min_img_size,
                                   const float4 *peaks, const int *counts, float4 *conns, int *conn_counts,
                                   unsigned *status, hipStream_t stream);
hipError_t launch_assemble(int batch, int maxp, int explicit_ids, const float4 *peaks, const int *counts,
                           const float4 *conns, const int *conn_counts, const unsigned *status, int flag_first,
                           pp_record *records, hipStream_t stream);

size_t lds_bytes_limb_py(int elem, int h, int w, int maxp, int cap);
size_t lds_bytes_assemble_py(int maxp);
hipError_t launch_limb_connect_py(const void *net, int dtype, int batch, int n_samples, int h, int w, int flip, int maxp,
                                  int cap, int img_height, const int *img_height_dev, const float4 *peaks, const int *counts,
                                  void *conns, int *conn_counts, unsigned *status, hipStream_t stream);
hipError_t launch_assemble_py(int batch, int maxp, int explicit_ids, const float4 *peaks, const int *counts, const void *conns,
                              const int *conn_counts, const unsigned *status, int flag_first, pp_record *records,
                              double *persons_out, int *n_persons_out, hipStream_t stream);
hipError_t launch_limb_connect_py_hwc(const float *paf, int H, int W, int C, int maxp, int cap, int img_height,
                                      const float4 *peaks, const int *counts, void *conns, int *conn_counts, unsigned *status,
                                      hipStream_t stream);

hipError_t launch_resize_cubic(const float *src, long src_plane, int src_ld, int ch, int cw, void *dst, int acc, int C, int dh,
                               int dw, double scale_x, double scale_y, float n_div, hipStream_t stream);
hipError_t launch_flip_average_planar(const void *net, int dtype, int batch, int h, int w, int flip, float *out,
                                      hipStream_t stream);
hipError_t launch_accumulate_scales(int n_scales, const void *const *nets, int dtype, int batch, const int *hs, const int *ws,
                                    int flip, const int *pad_down, const int *pad_right, int img_h, int img_w, double *heat_acc,
                                    double *paf_acc, hipStream_t stream);
hipError_t launch_fullres(int batch, int H, int W, float thre1, int maxp, int cap, int img_height, const double *heat_acc,
                          const double *paf_acc, unsigned char *mask_scratch, void *peaks64, int *counts, void *conns,
                          int *conn_counts, unsigned *status, pp_record *records, hipStream_t stream);

}  // namespace pp
#endif
