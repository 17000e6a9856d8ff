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

// x + (x moved by a DPP lane pattern): full-rate VALU, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    const int moved = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true);
    return x + __builtin_bit_cast(float, moved);
}
// sum over the 32 lanes of a half-wave, result in every lane: xor-1, xor-2 (quad_perm), half-mirror and
// mirror inside each row of 16, then one cross-row exchange
__device__ __forceinline__ float half_wave_sum(float x) {
    x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);  // row_half_mirror
    x = dpp_add<0x140>(x);  // row_mirror
    return x + __shfl_xor(x, 16);
}

// acc scale: the f16x3 path carries the 2^ws weight pre-scale in its accumulators
// read-modify-write store of one wave's tile (channel-major or time-major), EB rows per load batch
template <int MT, int NT, int EB, bool ACCUM>
__device__ __forceinline__ void conv_store_rmw(const ConvArgs& a, f32x16 (&acc)[MT][NT], float acc_scale, int b,
                                               int row0, int col0, int r, int h, int ncols, int stat_slot) {
    {
        // Latency-bound read-modify-write of the tile: issue the residual / accumulate loads of 8 rows x NT
        // columns back to back (addresses clamped so the loads need no branches), then combine and store.
        const bool tmaj = a.store == ST_TMAJOR;
        const bool has_res = a.resid != nullptr && !tmaj, has_bias = a.bias != nullptr;
        constexpr bool accum = ACCUM;
        const bool gelu = a.epi == EPI_GELU_NEW, do_div = a.out_div != 1.0f;
        const int cmax = ncols - 1, rmax = a.Cout - 1;
        // per-column element offsets (without the row term); merged mode maps column -> (utterance, t)
        long ycol[NT], rcol[NT];
        bool cok[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + nt * 32 + r;
            cok[nt] = col < ncols;
            const int cc = col < cmax ? col : cmax;
            int bb = b, tt = cc;
            if (a.merge_T > 0) {
                bb = cc / a.merge_T;
                tt = cc - bb * a.merge_T;
            }
            ycol[nt] = (long)bb * a.y_bs + (tmaj ? (long)tt * a.y_ld : (long)tt);
            rcol[nt] = (long)bb * a.r_bs + tt;
        }
        const long yrs = tmaj ? 1 : a.y_ld;  // row stride of the output
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int eg = 0; eg < 16 / EB; ++eg) {
                float rv[EB][NT], yv[ACCUM ? EB : 1][NT], bv[EB];
                long yo[EB];
                bool rok[EB];
#pragma unroll
                for (int e8 = 0; e8 < EB; ++e8) {
                    const int e = eg * EB + e8;
                    const int row = row0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    rok[e8] = row < a.Cout;
                    const int rowc = row < rmax ? row : rmax;
                    yo[e8] = (long)rowc * yrs;
                    bv[e8] = has_bias ? a.bias[rowc] : 0.f;
                    const long ro = (long)rowc * a.r_ld;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        rv[e8][nt] = has_res ? a.resid[ro + rcol[nt]] : 0.f;
                        if (ACCUM) yv[e8][nt] = a.y[yo[e8] + ycol[nt]];
                    }
                }
                float rs[EB], rq[EB];
#pragma unroll
                for (int e8 = 0; e8 < EB; ++e8) {
                    rs[e8] = 0.f;
                    rq[e8] = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = acc[mt][nt][eg * EB + e8] * acc_scale + bv[e8];
                        if (has_res) v += rv[e8][nt];
                        if (ACCUM) v += yv[e8][nt];
                        v *= a.out_mul;
                        if (do_div) v = v / a.out_div;
                        if (gelu) v = gelu_new_f(v);
                        if (rok[e8] && cok[nt]) a.y[yo[e8] + ycol[nt]] = v;
                        const float vm = cok[nt] ? v : 0.f;
                        rs[e8] += vm;
                        rq[e8] += vm * vm;
                    }
                }
                if (a.stat_part) {
                    // fused InstanceNorm statistics of what was just stored: reduce each row's partial over the
                    // 32 lanes of this half-wave (lanes = columns), one (sum, sumsq) per row and column slot
#pragma unroll
                    for (int e8 = 0; e8 < EB; ++e8) {
                        const float sv = half_wave_sum(rs[e8]), qv = half_wave_sum(rq[e8]);
                        if (r == 0 && rok[e8]) {
                            const int e = eg * EB + e8;
                            const int row = row0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            a.stat_part[((long)b * a.Cout + row) * a.stat_tiles + stat_slot] = make_float2(sv, qv);
                        }
                    }
                }
            }
        }
    }
}

// EB = rows per load batch of the read-modify-write path (8 or 16)
template <int MT, int NT, int EB = 8>
__device__ __forceinline__ void conv_store_tile(const ConvArgs& a, f32x16 (&acc)[MT][NT], float acc_scale, int b,
                                                int row0, int col0, int r, int h, int ncols, int Lout, int stat_slot) {
    if (a.store == ST_NORMAL || a.store == ST_TMAJOR) {
        if (a.accum != 0 && a.store == ST_NORMAL)
            conv_store_rmw<MT, NT, EB, true>(a, acc, acc_scale, b, row0, col0, r, h, ncols, stat_slot);
        else
            conv_store_rmw<MT, NT, EB, false>(a, acc, acc_scale, b, row0, col0, r, h, ncols, stat_slot);
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
