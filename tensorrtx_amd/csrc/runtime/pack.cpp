#include "pack.h"

#include <math.h>
#include <string.h>

#include "../kernels/kernels.h"

namespace trtx {

uint16_t f32_to_f16_bits(float f) {
    const _Float16 h = (_Float16)f;  // round-to-nearest-even, same as the device conversion
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

float f16_bits_to_f32(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

void pack_conv_weights_f16(const float* w, int cout, int cin, int kh, int kw, int cin_pad, int bk, const float* ch_scale,
                           uint16_t* packed) {
    const int bn = conv_igemm_pick_bn(cout);
    const int cout_pad = (cout + bn - 1) / bn * bn;
    const int kpad = (kh * kw * cin_pad + bk - 1) / bk * bk;  // cin_pad = per-tap K stride (CinK); rows padded to the k-step
    memset(packed, 0, sizeof(uint16_t) * (size_t)cout_pad * kpad);
    for (int co = 0; co < cout; ++co) {
        const float sc = ch_scale ? ch_scale[co] : 1.0f;
        for (int c = 0; c < cin; ++c)
            for (int r = 0; r < kh; ++r)
                for (int q = 0; q < kw; ++q) {
                    const float v = w[(((size_t)co * cin + c) * kh + r) * kw + q] * sc;
                    packed[(size_t)co * kpad + (size_t)(r * kw + q) * cin_pad + c] = f32_to_f16_bits(v);
                }
    }
}

}  // namespace trtx

// kINT8: per-output-channel symmetric quantisation of the (BN-folded) weights, rows [Cout_pad][Kpad] with k = tap * cink + c
void trtx::conv_pack_weights_i8(const float* w, int cout, int cin, int kh, int kw, int cink, const float* ch_scale, int cout_pad, int kpad,
                                int8_t* packed, float* wscale_out) {
    memset(packed, 0, (size_t)cout_pad * kpad);
    for (int co = 0; co < cout_pad; ++co) wscale_out[co] = 1.0f;
    for (int co = 0; co < cout; ++co) {
        const float sc = ch_scale ? ch_scale[co] : 1.0f;
        float amax = 0.f;
        const size_t n = (size_t)cin * kh * kw;
        for (size_t i = 0; i < n; ++i) {
            const float v = fabsf(w[(size_t)co * n + i] * sc);
            amax = v > amax ? v : amax;
        }
        const float s = amax > 0.f ? amax / 127.0f : 1.0f;
        wscale_out[co] = s;
        for (int c = 0; c < cin; ++c)
            for (int r = 0; r < kh; ++r)
                for (int q = 0; q < kw; ++q) {
                    float t = nearbyintf(w[(((size_t)co * cin + c) * kh + r) * kw + q] * sc / s);
                    t = t > 127.f ? 127.f : (t < -127.f ? -127.f : t);
                    packed[(size_t)co * kpad + (size_t)(r * kw + q) * cink + c] = (int8_t)t;
                }
    }
}

namespace trtx {

void pack_conv_weights_f32(const float* w, int cout, int cin_g, int kh, int kw, const float* ch_scale, float* packed) {
    for (int co = 0; co < cout; ++co) {
        const float sc = ch_scale ? ch_scale[co] : 1.0f;
        for (int c = 0; c < cin_g; ++c)
            for (int r = 0; r < kh; ++r)
                for (int q = 0; q < kw; ++q)
                    packed[(((size_t)co * kh + r) * kw + q) * cin_g + c] =
                            w[(((size_t)co * cin_g + c) * kh + r) * kw + q] * sc;
    }
}

void pack_deconv_weights_f32(const float* w, int cin, int cout, int groups, int kh, int kw, float* packed) {
    const int cin_g = cin / groups, cout_g = cout / groups;
    for (int co = 0; co < cout; ++co) {
        const int g = co / cout_g, col = co % cout_g;
        for (int c = 0; c < cin_g; ++c) {
            const int ci = g * cin_g + c;
            for (int r = 0; r < kh; ++r)
                for (int q = 0; q < kw; ++q)
                    packed[(((size_t)co * kh + r) * kw + q) * cin_g + c] =
                            w[(((size_t)ci * cout_g + col) * kh + r) * kw + q];
        }
    }
}

}  // namespace trtx
