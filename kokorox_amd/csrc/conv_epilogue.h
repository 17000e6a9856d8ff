// Shared epilogue of the two conv1d MFMA kernels (conv_mfma.hip, conv_f16x3.hip).
// The store form is chosen ONCE per workgroup (wave-uniform branch), so the per-element code of the
// common channel-major path stays a handful of instructions.
// C/D map of every 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma once
#include "kx_common.h"

namespace kx {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float gelu_new_f(float x) {
    const float c = 0.7978845608028654f;
    return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * (x * x * x))));
}

// acc scale: the f16x3 path carries the 2^ws weight pre-scale in its accumulators
template <int MT, int NT>
__device__ __forceinline__ void conv_store_tile(const ConvArgs& a, f32x16 (&acc)[MT][NT], float acc_scale, int b,
                                                int row0, int col0, int r, int h, int ncols, int Lout) {
    if (a.store == ST_NORMAL) {
        const bool has_res = a.resid != nullptr, has_bias = a.bias != nullptr;
        const bool accum = a.accum != 0, gelu = a.epi == EPI_GELU_NEW, do_div = a.out_div != 1.0f;
        float* yb = a.y + (long)b * a.y_bs;
        const float* rb = has_res ? a.resid + (long)b * a.r_bs : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= a.Cout) continue;
                const float bv = has_bias ? a.bias[row] : 0.f;
                float* yr = yb + (long)row * a.y_ld;
                const float* rr = has_res ? rb + (long)row * a.r_ld : nullptr;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = col0 + nt * 32 + r;
                    if (col >= ncols) continue;
                    float v = acc[mt][nt][e] * acc_scale + bv;
                    if (has_res) v += rr[col];
                    if (accum) v += yr[col];
                    v *= a.out_mul;
                    if (do_div) v = v / a.out_div;
                    if (gelu) v = gelu_new_f(v);
                    yr[col] = v;
                }
            }
        }
        return;
    }
    if (a.store == ST_TMAJOR) {  // [B][L][y_ld]: LSTM gate pre-activations (bias only)
        float* yb = a.y + (long)b * a.y_bs;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= a.Cout) continue;
                const float bv = a.bias ? a.bias[row] : 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = col0 + nt * 32 + r;
                    if (col >= ncols) continue;
                    yb[(long)col * a.y_ld + row] = (acc[mt][nt][e] * acc_scale + bv) * a.out_mul;
                }
            }
        }
        return;
    }
    // ST_UPSCATTER: polyphase transposed conv, GEMM row (p, co), column q -> out[co][s*q + p - pad]
    float* yb = a.y + (long)b * a.y_bs;
    const float* rb = a.resid ? a.resid + (long)b * a.r_bs : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = row0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row >= a.Cout) continue;
            const int p = row / a.up_cout;
            const int co = row - p * a.up_cout;
            const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = col0 + nt * 32 + r;
                if (col >= ncols) continue;
                const int tout = a.up_s * col + p - a.up_pad;
                if (tout < 0 || tout >= Lout) continue;
                const float v = acc[mt][nt][e] * acc_scale + bv;
                const long off = (long)co * a.y_ld + tout + a.up_off;
                float o = v;
                if (rb) o += rb[(long)co * a.r_ld + tout + a.up_off];
                yb[off] = o;
                if (a.up_reflect && tout == 1) {  // ReflectionPad1d((1,0)): out[0] = up[1]
                    float o0 = v;
                    if (rb) o0 += rb[(long)co * a.r_ld];
                    yb[(long)co * a.y_ld] = o0;
                }
            }
        }
    }
}

}  // namespace kx
