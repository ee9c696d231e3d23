// Host-side weight re-layout used at engine-build time (and by the single-op test entry points).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace trtx {

// TensorRT conv weights are KCRS fp32 ([Cout][Cin][kh][kw]).  The implicit-GEMM kernel wants
// fp16 [Cout_pad][Kpad = kh*kw*cin_pad rounded up to bk] with k = (r*kw + q)*cin_pad + c (cin_pad = CinK, the per-tap stride), zero padded; an optional per-output-channel
// scale (folded BatchNorm, yolov8/src/block.cpp:45-77) is multiplied in before rounding to fp16.
void pack_conv_weights_f16(const float* w_kcrs, int cout, int cin, int kh, int kw, int cin_pad, int bk,
                           const float* ch_scale, uint16_t* packed);

// fp32 [Cout][kh][kw][Cin/groups] for the generic direct kernel (KCRS source, Cin here = per group)
void pack_conv_weights_f32(const float* w_kcrs, int cout, int cin_g, int kh, int kw, const float* ch_scale,
                           float* packed);

// TensorRT deconvolution weights are CKRS ([Cin][Cout/groups][kh][kw]); gather form for
// deconv_direct is [Cout][kh][kw][Cin/groups].
void pack_deconv_weights_f32(const float* w_ckrs, int cin, int cout, int groups, int kh, int kw, float* packed);

uint16_t f32_to_f16_bits(float f);
float f16_bits_to_f32(uint16_t h);

}  // namespace trtx
