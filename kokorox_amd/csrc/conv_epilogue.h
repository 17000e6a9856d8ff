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
        // Latency-bound read-modify-write of the tile: issue the residual / accumulate loads of 8 rows x NT
        // columns back to back (addresses clamped so the loads need no branches), then combine and store.
        const bool has_res = a.resid != nullptr, has_bias = a.bias != nullptr;
        const bool accum = a.accum != 0, gelu = a.epi == EPI_GELU_NEW, do_div = a.out_div != 1.0f;
        float* yb = a.y + (long)b * a.y_bs;
        const float* rb = has_res ? a.resid + (long)b * a.r_bs : yb;
        const int cmax = ncols - 1, rmax = a.Cout - 1;
        int colc[NT];
        bool cok[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + nt * 32 + r;
            cok[nt] = col < ncols;
            colc[nt] = col < cmax ? col : cmax;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int eg = 0; eg < 2; ++eg) {
                float rv[8][NT], yv[8][NT], bv[8];
                long yo[8];
                bool rok[8];
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = eg * 8 + e8;
                    const int row = row0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    rok[e8] = row < a.Cout;
                    const int rowc = row < rmax ? row : rmax;
                    yo[e8] = (long)rowc * a.y_ld;
                    bv[e8] = has_bias ? a.bias[rowc] : 0.f;
                    const long ro = (long)rowc * (has_res ? a.r_ld : a.y_ld);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        rv[e8][nt] = has_res ? rb[ro + colc[nt]] : 0.f;
                        yv[e8][nt] = accum ? yb[yo[e8] + colc[nt]] : 0.f;
                    }
                }
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = acc[mt][nt][eg * 8 + e8] * acc_scale + bv[e8];
                        if (has_res) v += rv[e8][nt];
                        if (accum) v += yv[e8][nt];
                        v *= a.out_mul;
                        if (do_div) v = v / a.out_div;
                        if (gelu) v = gelu_new_f(v);
                        if (rok[e8] && cok[nt]) yb[yo[e8] + colc[nt]] = v;
                    }
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
