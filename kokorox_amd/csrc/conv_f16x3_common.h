// Device helpers shared by the f16x3 conv kernels (conv_f16x3.hip, conv_f16x3_da.hip, conv_f16x3_dag.hip): the split-f16 operand
// format, the snake / leaky input activation and the asynchronous global -> LDS copy.
#pragma once
#include "conv_epilogue.h"
#include "kx_common.h"

namespace kx {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;

#ifndef KX_EPI_ROWS
#define KX_EPI_ROWS 4
#endif
constexpr int EPI_ROWS = KX_EPI_ROWS;  // rows per epilogue load batch
constexpr int CK16 = 16;  // input channels per K-chunk

// sin^2(t) for moderate |t| (snake activations).  sin^2 has period pi and is even about every multiple of it, so
// one Cody-Waite reduction to r = t - n pi, |r| <= pi/2, and one even polynomial do the whole job, with no quadrant
// select: sin^2(r) = z P(z), z = r^2, P a degree-5 near-minimax fit on [0, (pi/2)^2] (approximation error 2.5e-8;
// 1.6e-7 worst case as evaluated in f32, tools/probes/sin_accuracy.hip).  14 vector instructions with the
// surrounding alpha multiply and the final fma, against 19 for the quarter-period form with its parity select.
__device__ __forceinline__ float sin_sq(float t) {
    const float n = rintf(t * 0.318309886183790672f);          // t / pi
    float r = fmaf(n, -3.14159274101257324f, t);                // pi, high part (the f32 nearest pi)
    r = fmaf(n, 8.74227765734758578e-08f, r);                   // minus the low part (pi - hi = -8.74e-8)
    const float z = r * r;
    float p = fmaf(z, -3.6197402550897095e-06f, 1.3928599946666651e-04f);
    p = fmaf(z, p, -3.1722760759294033e-03f);
    p = fmaf(z, p, 4.4443082064390182e-02f);
    p = fmaf(z, p, -3.3333304524421692e-01f);
    p = fmaf(z, p, 1.0f);
    return z * p;
}

// f32 -> (hi, lo) halves, two values packed per dword: hi = f16(v), lo = f16(v - hi)
using float2v = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi_pk, unsigned& lo_pk) {
    float2v v;
    v[0] = __builtin_amdgcn_fmed3f(v0, -65504.f, 65504.f);
    v[1] = __builtin_amdgcn_fmed3f(v1, -65504.f, 65504.f);
    const half2v h2 = __builtin_convertvector(v, half2v);          // one v_cvt_pk_f16_f32
    // (the two residuals as SCALAR subtractions: written as one float2 expression they become a v_pk_add_f32, and packed f32
    // vector instructions are slow beside MFMAs on gfx950 -- MI355X_MICROARCH.md, "price of one filler beside MFMAs")
    float2v d;
    d[0] = v[0] - (float)h2[0];
    d[1] = v[1] - (float)h2[1];
    asm volatile("" : "+v"(d[0]));  // (keeps the SLP vectoriser from re-packing the pair)
    const half2v l2 = __builtin_convertvector(d, half2v);
    hi_pk = __builtin_bit_cast(unsigned, h2);
    lo_pk = __builtin_bit_cast(unsigned, l2);
}

// f16f8 mode: hi_pk as split_pair; x_pk = the pair's 8-bit operands of the two cross terms of a product,
// [e4m3(v0), e4m3(v1), e4m3(2^11 (v0 - hi0)), e4m3(2^11 (v1 - hi1))], each clamped to e4m3's finite range (the conversion
// itself does not saturate).  2^11: a residual is at most half an ulp of its f16 high part, so for |v| in [2^-6, 448] both
// images lie in e4m3's normal range; outside it the cross terms lose bits, never the product's leading term.
__device__ __forceinline__ void split_pair_f8(float v0, float v1, unsigned& hi_pk, unsigned& x_pk) {
    float2v v;
    v[0] = __builtin_amdgcn_fmed3f(v0, -65504.f, 65504.f);
    v[1] = __builtin_amdgcn_fmed3f(v1, -65504.f, 65504.f);
    const half2v h2 = __builtin_convertvector(v, half2v);
    float d0 = (v[0] - (float)h2[0]) * 2048.f;
    float d1 = (v[1] - (float)h2[1]) * 2048.f;
    asm volatile("" : "+v"(d0));  // (keeps the SLP vectoriser from packing the pair: see split_pair)
    const float c0 = __builtin_amdgcn_fmed3f(v0, -448.f, 448.f), c1 = __builtin_amdgcn_fmed3f(v1, -448.f, 448.f);
    d0 = __builtin_amdgcn_fmed3f(d0, -448.f, 448.f);
    d1 = __builtin_amdgcn_fmed3f(d1, -448.f, 448.f);
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(c0, c1, 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(d0, d1, pk, true);
    hi_pk = __builtin_bit_cast(unsigned, h2);
    x_pk = (unsigned)pk;
}

template <int ACT>
__device__ __forceinline__ float in_act(float y, float slope, float al, float ial) {
    if (ACT == ACT_SNAKE) return fmaf(ial, sin_sq(al * y), y);
    if (ACT == ACT_LEAKY) return y > 0.f ? y : y * slope;
    return y;
}

// one 1-KiB wave-instruction of an asynchronous global -> LDS copy (lane i moves 16 B to base + 16 i)
__device__ __forceinline__ void glds16(const uint4* gsrc_lane, uint4* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

}  // namespace kx
