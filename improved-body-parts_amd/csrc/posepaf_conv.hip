// C ABI of the fused convolution (include/posepaf.h: pp_conv_f16): argument checks + dispatch to one of the tile
// configurations compiled from posepaf_conv_inst.hip.
#include <cstdint>
#include <hip/hip_runtime.h>

#include "../../include/posepaf.h"
#include "posepaf_conv.h"

extern "C" int pp_conv_num_configs(void) { return kNumConvConfigs; }

extern "C" int pp_conv_ld_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd,
                              int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int config, int ldx,
                              int ldy, void *stream);

extern "C" int pp_conv_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd,
                           int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int config,
                           void *stream) {
    return pp_conv_ld_f16(x, w, bias, extra, y, n, h, wd, c_in, c_out, ksize, pad, dilation, extra_mode, slope, config, c_in, c_out, stream);
}

// the same with explicit pixel strides of x and y (elements; >= the channel counts, multiples of 8): either may be a channel
// slice of a wider NHWC tensor
extern "C" int pp_conv_ld_f16(const void *x, const void *w, const void *bias, const void *extra, void *y, int n, int h, int wd,
                              int c_in, int c_out, int ksize, int pad, int dilation, int extra_mode, float slope, int config, int ldx,
                              int ldy, void *stream) {
    if (ldx < c_in || ldy < c_out || (ldx & 7) || (ldy & 7)) return PP_ERR_BAD_ARG;
    if (!x || !w || !bias || !y || n <= 0 || h <= 0 || wd <= 0 || c_in <= 0 || c_out <= 0 || ksize <= 0 || pad < 0 ||
        dilation <= 0 || extra_mode < 0 || extra_mode > 2 || (extra_mode != 0) != (extra != nullptr))
        return PP_ERR_BAD_ARG;
    if (config < 0 || config >= kNumConvConfigs) return PP_ERR_BAD_ARG;
    if ((c_in & 7) || (c_out & 7)) return PP_ERR_UNSUPPORTED;  // 16-byte vector loads/stores along the channel axis
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(bias) |
                         reinterpret_cast<uintptr_t>(extra) | reinterpret_cast<uintptr_t>(y);
    if (al & 15) return PP_ERR_BAD_ARG;
    const long ho = (long)h + 2L * pad - (long)dilation * (ksize - 1), wo = (long)wd + 2L * pad - (long)dilation * (ksize - 1);
    if (ho <= 0 || wo <= 0) return PP_ERR_BAD_ARG;
    // 32-bit byte offsets inside the kernels: each tensor at most 2 GB
    if ((long)n * h * wd * ldx * 2 > (1L << 31) || (long)n * ho * wo * ldy * 2 > (1L << 31)) return PP_ERR_TOO_LARGE;
    const PPConvArgs a{x, w, bias, extra, y, n, h, wd, c_in, c_out, ksize, ksize, pad, dilation, ldx, ldy, extra_mode, slope, stream};
    int rc = -1;
    switch (config) {
        case 0: rc = pp_conv_run_cfg0(a); break;
        case 1: rc = pp_conv_run_cfg1(a); break;
        case 2: rc = pp_conv_run_cfg2(a); break;
        case 3: rc = pp_conv_run_cfg3(a); break;
        case 4: rc = pp_conv_run_cfg4(a); break;
        case 5: rc = pp_conv_run_cfg5(a); break;
        case 6: rc = pp_conv_run_cfg6(a); break;
        case 7: rc = pp_conv_run_cfg7(a); break;
        case 8: rc = pp_conv_run_cfg8(a); break;
        case 9: rc = pp_conv_run_cfg9(a); break;
    }
    if (rc != 0) return PP_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
}
