// Internal: arguments of one fused convolution launch (csrc/posepaf_conv_inst.hip, csrc/posepaf_conv.hip).
#pragma once
struct PPConvArgs {
    const void *x, *w, *bias, *extra;  // NHWC fp16, (K, R, S, C) fp16, fp16[K], optional NHWC (N, Ho, Wo, K) fp16
    void *y;                           // (N, Ho, Wo, K) fp16
    int N, H, W, C, K, R, S, pad, dil;
    int ldx, ldy;                      // elements between consecutive pixels of x / y (C / K: packed; larger: a channel slice of a wider
                                       // NHWC tensor -- the backbone's concatenation, models/layers_transposed.py:193-195)
    int extra_mode;                    // 0 none, 1 added before the activation (residual), 2 after it (post)
    float slope;                       // LeakyReLU slope; 1.0 = no activation
    void *stream;
};
constexpr int kNumConvConfigs = 10;
// returns 0 ok, -1 shape not supported by this tile configuration
#define PP_CONV_DECL(n) int pp_conv_run_cfg##n(const PPConvArgs &a);
PP_CONV_DECL(0) PP_CONV_DECL(1) PP_CONV_DECL(2) PP_CONV_DECL(3) PP_CONV_DECL(4) PP_CONV_DECL(5) PP_CONV_DECL(6) PP_CONV_DECL(7)
PP_CONV_DECL(8) PP_CONV_DECL(9)
#undef PP_CONV_DECL
