// Shared epilogue of the two conv1d MFMA kernels (conv_mfma.hip, conv_f16x3.hip).
// The store form is chosen ONCE per workgroup (wave-uniform branch), so the per-element code of the
// common channel-major path stays a handful of instructions.
// C/D map of every 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma once
#include "kx_common.h"

// epilogue load batches (EB rows x NT tiles) in flight ahead of the stores; "SMALL" = waves with fewer than 8 accumulator
// tiles, which run at three workgroups per CU under a 168-register cap
#ifndef KX_EPI_DEPTH_R
#define KX_EPI_DEPTH_R 2  // residual form (deeper queues were measured: 5 = no gain, 7 = slower)
#endif
#ifndef KX_EPI_DEPTH_A
#define KX_EPI_DEPTH_A 1  // residual + running-sum form (two loads per element)
#endif
#ifndef KX_EPI_DEPTH_R_SMALL
#define KX_EPI_DEPTH_R_SMALL 2
#endif
#ifndef KX_EPI_DEPTH_A_SMALL
#define KX_EPI_DEPTH_A_SMALL 1
#endif

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
// sum over the 32 lanes of a half-wave, all on the vector unit: xor-1, xor-2 (quad_perm), half-mirror and mirror
// inside each row of 16, then row_bcast15 (lane 15 of rows 0 / 2 added into every lane of rows 1 / 3).  The total is
// valid in the UPPER 16 lanes of each half-wave (lanes 16-31 and 48-63).
constexpr int HALF_WAVE_SUM_LANE = 16;  // r = lane & 31 of a lane that holds the total
__device__ __forceinline__ float half_wave_sum(float x) {
    x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);  // row_half_mirror
    x = dpp_add<0x140>(x);  // row_mirror
    const int up = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false);  // row_bcast15
    return x + __builtin_bit_cast(float, up);
}

// v / d for a launch-constant divisor, rd = 1 / d: one Newton step on the product makes the quotient correctly
// rounded except in rare double-rounding cases (<= 1 ulp), at 3 VALU instead of the ~10 of a full division
__device__ __forceinline__ float div_const(float v, float d, float rd) {
    const float q = v * rd;
    return fmaf(fmaf(-q, d, v), rd, q);
}

// raw buffer access: address = descriptor base + per-lane byte offset (VGPR) + wave-uniform byte offset (SGPR),
// one instruction per access and no 64-bit address arithmetic on the vector unit
using buf_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ buf_rsrc make_buf(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);  // raw, 4 GiB window
}
// cache policy of the epilogue's accesses (aux operand: 1 = sc0, 2 = nt, 4 = sc1)
#ifndef KX_EPI_LD_AUX
#define KX_EPI_LD_AUX 0
#endif
#ifndef KX_EPI_ST_AUX
#define KX_EPI_ST_AUX 0
#endif
__device__ __forceinline__ float buf_load(buf_rsrc rs, unsigned lane_bytes, unsigned uniform_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_bytes, uniform_bytes, KX_EPI_LD_AUX));
}
__device__ __forceinline__ void buf_store(buf_rsrc rs, unsigned lane_bytes, unsigned uniform_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, lane_bytes, uniform_bytes, KX_EPI_ST_AUX);
}

// Store of one wave's MT x NT block of 32x32 accumulator tiles, channel-major or time-major, with bias, residual,
// accumulate, scale / divide, gelu_new and the fused InstanceNorm partial sums.  The accumulators of the f16x3
// path carry the 2^ws weight pre-scale (acc_scale undoes it).
//   FULL  : the block lies wholly inside [0, Cout) x [0, ncols): no predicates at all (the common case).
//   !FULL : edge blocks; every store is predicated, load rows / columns are clamped.
// Addressing: one buffer descriptor per tensor and utterance, a per-lane byte offset that does not depend on the
// row, and a wave-uniform row term in a scalar register (FULL; the MFMA C/D map puts row + 4 in the upper
// half-wave, which is folded into the lane offset).  Offsets are 32-bit: a lane never reaches past its own
// utterance's tensor (merged mode: the one token-axis tensor), which the launchers keep below 4 GiB.
//   LOADS : 0 = nothing to read back, 1 = residual, 2 = residual + accumulate into y, 3 = decided at run time
//           (edge blocks only; the compile-time forms keep the common path free of per-row branches)
template <int MT, int NT, int EB, int LOADS, bool FULL, bool GELU>
__device__ __forceinline__ void conv_store_rmw(const ConvArgs& a, f32x16 (&acc)[MT][NT], float acc_scale, int b,
                                               int row0, int col0, int r, int h, int ncols, int stat_slot,
                                               float2* stat_scr) {
    const bool tmaj = a.store == ST_TMAJOR;
    const bool has_bias = a.bias != nullptr;
    const bool has_res = LOADS == 3 ? (a.resid != nullptr && !tmaj) : (LOADS >= 1);
    const bool accum = LOADS == 3 ? (a.accum != 0 && !tmaj) : (LOADS == 2);
    constexpr bool ACCUM = LOADS >= 2;  // (whether the y read-back buffers exist)
    const int cmax = ncols - 1, rmax = a.Cout - 1;
    const bool merged = a.merge_T > 0;
    // row strides in bytes (timing ablations, results wrong: KX_DBG bit 32 stores every row of the tensor over row 0,
    // bit 64 reads the residual / running sum from row 0 -- same instruction stream, no HBM traffic behind it)
    const unsigned yrs = (a.dbg & 32) ? 0u : 4u * (tmaj ? 1u : (unsigned)a.y_ld);
    const unsigned rrs = (a.dbg & 64) ? 0u : 4u * (unsigned)a.r_ld;
    const buf_rsrc ybuf = make_buf(a.y + (merged ? 0 : (long)b * a.y_bs));
    const buf_rsrc rbuf = make_buf(has_res ? a.resid + (merged ? 0 : (long)b * a.r_bs) : a.y);
    // bias of the wave's MT * 32 <= 64 rows: one value per lane, handed out below with v_readlane
    const int brow = row0 + r + 32 * h;
    const float bias_lane = has_bias ? a.bias[brow < rmax ? brow : rmax] : 0.f;
    unsigned yoff[NT], roff[NT];
    bool cok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = col0 + nt * 32 + r;
        cok[nt] = FULL || col < ncols;
        const int cc = FULL ? col : (col < cmax ? col : cmax);
        int bb = 0, tt = cc;
        if (merged) {
            bb = cc / a.merge_T;
            tt = cc - bb * a.merge_T;
        }
        yoff[nt] = 4u * ((unsigned)((long)bb * a.y_bs) + (tmaj ? (unsigned)tt * (unsigned)a.y_ld : (unsigned)tt)) +
                   (FULL ? (unsigned)(4 * h) * yrs : 0u);
        roff[nt] = 4u * ((unsigned)((long)bb * a.r_bs) + (unsigned)tt) + (FULL ? (unsigned)(4 * h) * rrs : 0u);
    }
    // row of the h = 0 lanes of accumulator register e in tile mt (wave-uniform); upper half-wave: + 4
    auto row_of = [&](int mt, int e) { return row0 + mt * 32 + (e & 3) + 8 * (e >> 2); };
    // FULL: the half-wave's 4-row step is already in the lane offsets, the row term is wave-uniform (scalar).
    // !FULL: per-lane row, clamped to the last valid one (loads stay in bounds, stores are masked).
    auto row_term = [&](int rowu, unsigned stride_bytes) -> unsigned {
        if (FULL) return (unsigned)rowu * stride_bytes;
        const int rl = rowu + 4 * h;
        return (unsigned)(rl < rmax ? rl : rmax) * stride_bytes;
    };
    // (FULL: lane offset in the VGPR operand, row term in the SGPR operand; !FULL: both per lane)
    auto ldv = [&](buf_rsrc rs, unsigned lane_off, unsigned rt) {
        return FULL ? buf_load(rs, lane_off, rt) : buf_load(rs, lane_off + rt, 0u);
    };
    auto stv = [&](buf_rsrc rs, unsigned lane_off, unsigned rt, float v) {
        if (FULL) buf_store(rs, lane_off, rt, v);
        else buf_store(rs, lane_off + rt, 0u, v);
    };

    // gfx950 retires vector loads and stores through ONE in-order counter (vmcnt): a load issued after a store
    // cannot be waited for without also waiting for that store's write acknowledgement, which turns a plain
    // load / combine / store loop into a chain of round trips.  So the loads of batch i+1 are issued BEFORE the
    // stores of batch i (bias comes from a register, see above), and scheduling barriers keep the compiler from
    // re-bunching the batches (which it otherwise does, spilling what the loads return).
    constexpr int NEG = 16 / EB;     // load batches per 32-row tile
    constexpr int NBAT = MT * NEG;   // batches of EB rows x NT columns
    const bool has_loads = has_res || accum;
    // the divisor (mean over the resblock kernels) only ever comes with the accumulate form
    const bool do_div = (LOADS == 2 || LOADS == 3) && a.out_div != 1.0f;
    const float div_d = a.out_div, div_rd = 1.0f / a.out_div;
    const bool gelu = GELU && a.epi == EPI_GELU_NEW;
    const bool want_stats = a.stat_part != nullptr;
    // load batches in flight ahead of the one being stored.  The epilogue of a tile is a chain of memory round trips
    // (a batch is stored only after its residual / running-sum values are back, ~2 us under this kernel's own load), so
    // its length is (batches / DEPTH) x latency; the staging registers of the main loop are dead here, which pays for
    // the deeper queues of the compile-time forms.
    constexpr bool BIG = MT * NT >= 8;
    constexpr int DEPTH = LOADS == 1 ? (BIG ? KX_EPI_DEPTH_R : KX_EPI_DEPTH_R_SMALL)
                        : LOADS == 2 ? (BIG ? KX_EPI_DEPTH_A : KX_EPI_DEPTH_A_SMALL) : (ACCUM ? 1 : 2);
    constexpr int NBUF = DEPTH + 1;
    float rv[NBUF][EB][NT], yv[NBUF][ACCUM ? EB : 1][NT];
    auto load_batch = [&](int bi, float (&rvb)[EB][NT], float (&yvb)[ACCUM ? EB : 1][NT]) {
        const int mt = bi / NEG, eg = bi % NEG;
#pragma unroll
        for (int e8 = 0; e8 < EB; ++e8) {
            const int rowu = row_of(mt, eg * EB + e8);
            const unsigned rt = row_term(rowu, rrs), yt = row_term(rowu, yrs);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (LOADS == 3) {
                    rvb[e8][nt] = has_res ? ldv(rbuf, roff[nt], rt) : 0.f;
                    yvb[e8][nt] = accum ? ldv(ybuf, yoff[nt], yt) : 0.f;
                } else {
                    rvb[e8][nt] = ldv(rbuf, roff[nt], rt);
                    if (ACCUM) yvb[e8][nt] = ldv(ybuf, yoff[nt], yt);
                }
            }
        }
    };
    if (LOADS != 0 && has_loads) {
#pragma unroll
        for (int d = 0; d < DEPTH && d < NBAT; ++d) load_batch(d, rv[d], yv[d]);
    }
#pragma unroll
    for (int bi = 0; bi < NBAT; ++bi) {
        const int mt = bi / NEG, eg = bi % NEG;
        if (LOADS != 0 && bi + DEPTH < NBAT && has_loads)
            load_batch(bi + DEPTH, rv[(bi + DEPTH) % NBUF], yv[(bi + DEPTH) % NBUF]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e8 = 0; e8 < EB; ++e8) {
            const int e = eg * EB + e8;
            const int rowu = row_of(mt, e);
            const bool rok = FULL || rowu + 4 * h <= rmax;
            const unsigned yt = row_term(rowu, yrs);  // (clamped rows are exactly the masked ones)
            // this lane's row is tile row c (lower half-wave) or c + 4 (upper), c a compile-time constant
            const int c = mt * 32 + (e & 3) + 8 * (e >> 2);
            const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bias_lane), c));
            const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bias_lane), c + 4));
            const float bvv = h ? b1 : b0;
            float rs = 0.f, rq = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                // (every multiply-add of the epilogue is an explicit fma: left to -ffp-contract, the same source fused or
                // not depending on how the surrounding kernel was vectorised, and batch members then differed from their
                // batch-of-one runs in the last bit of the statistics)
                float v = __builtin_fmaf(acc[mt][nt][e], acc_scale, bvv);
                if (LOADS == 3) {
                    if (has_loads) v += rv[bi % NBUF][e8][nt] + yv[bi % NBUF][e8][nt];
                } else {
                    if (LOADS >= 1) v += rv[bi % NBUF][e8][nt];
                    if (LOADS == 2) v += yv[bi % NBUF][e8][nt];
                }
                v *= a.out_mul;
                if (LOADS == 2) v = div_const(v, div_d, div_rd);  // (exact no-op for a divisor of 1)
                if (LOADS == 3 && do_div) v = div_const(v, div_d, div_rd);
                if (gelu) v = gelu_new_f(v);
                if (FULL || (rok && cok[nt])) stv(ybuf, yoff[nt], yt, v);
                const float vm = (FULL || cok[nt]) ? v : 0.f;
                rs += vm;
                rq = __builtin_fmaf(vm, vm, rq);
            }
            if (want_stats && stat_scr != nullptr) {
                // fused InstanceNorm statistics of what was just stored, LDS form: every lane parks the partial of its row
                // (its column of the NT tiles) in the wave's scratch [32 tile rows][32 columns + 1]; when the tile's 32
                // rows are complete, lane l < 32 adds up row l.  One ds_write_b64 per register row and 32 ds_read_b64 +
                // 64 adds per tile, against a 5-step DPP reduction of two values per register row (measured: the
                // reductions were ~2/3 of the epilogue's instructions, and the epilogue 11-14 % of the kernel).
                // LDS operations of one wave execute in order, so no barrier is involved.
                stat_scr[((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = make_float2(rs, rq);
                if (eg == NEG - 1 && e8 == EB - 1 && h == 0) {
                    float2 t = stat_scr[r * 33];
#pragma unroll
                    for (int j = 1; j < 32; ++j) {
                        const float2 pj = stat_scr[r * 33 + j];
                        t.x += pj.x;
                        t.y += pj.y;
                    }
                    const int rowg = row0 + mt * 32 + r;
                    if (FULL || rowg <= rmax) a.stat_part[((long)b * a.Cout + rowg) * a.stat_tiles + stat_slot] = t;
                }
            } else if (want_stats) {
                // DPP form (kernels without LDS to spare): reduce the row's partial over the 32 lanes of this half-wave
                // (lanes = columns), one (sum, sumsq) per row and column slot
                const float sv = half_wave_sum(rs), qv = half_wave_sum(rq);
                if (r == HALF_WAVE_SUM_LANE && rok)
                    a.stat_part[((long)b * a.Cout + rowu + 4 * h) * a.stat_tiles + stat_slot] = make_float2(sv, qv);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- wide form of the read-modify-write store (direct-A kernels, interior 32-row x 128-column groups) ----------------------
// Measured (profiles/r03_epi_pattern_probe.txt): the dword form above issues 2 rows x 128 B per instruction; the same bytes
// moved as whole 128-byte lines, 16 B per lane (8 rows x 128 B per instruction), issue 1.75x faster and halve what the
// epilogue exposes beside MFMA work.  A lane cannot do that from the MFMA C/D layout (it holds ONE column of a tile), so
// every 32 x 32 tile goes through a 4 KiB LDS transpose: 16 ds_write_b32 (register e = rows (e&3)+8(e>>2)+4h, one column per
// lane: 2 rows x 32 floats per instruction, conflict-free at a pitch of 32 floats), then 4 ds_read_b128 in the store layout:
// lane' = (row 8 i + (lane >> 3), columns 4 (lane & 7) .. + 3), also conflict-free.  LDS operations of one wave execute in
// order, so the scratch needs no barrier.  Residual / running-sum loads use the same layout and need no transpose; they are
// issued one unit (two tiles; one when there are two loads per element) ahead of the stores, which come later in the wave's
// in-order memory queue.  Per element the arithmetic is that of conv_store_rmw, operation for operation (the stored values
// are bit-identical); the InstanceNorm partial sums are added in another order (4 columns per lane, 8 lanes per row, DPP),
// which both tile widths of the direct-A kernel share (batch invariance).
// SC: scatter(n) writes the wave's 32 x 32 tile n of the group into scr, element (row, column) at scr[row * P + column]
// (P: pitch in floats, a multiple of 4) -- the only part that knows the MFMA's C/D layout.
// GT: 32-column tiles of the group (4 = 128 columns; 2 = the 64-column groups of the S16 form)
// AUXV: cache policy of the residual / running-sum loads and of the stores (2 = non-temporal: ConvArgs::epi_stream)
template <int LOADS, int P, class SC, int GT = 4, int AUXV = 0>
__device__ __forceinline__ void conv_store_wide4_t(const ConvArgs& a, SC&& scatter, float acc_scale, int b, int row0, int col0,
                                                   int lane, int stat_slot, float* scr) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int rr = lane >> 3, pc = lane & 7;  // store layout: row 8 i + rr of the wave's 32, 16-byte piece pc of a tile row
    const bool want_stats = a.stat_part != nullptr;
    const buf_rsrc ybuf = make_buf(a.y + (long)b * a.y_bs);
    const buf_rsrc rbuf = make_buf(LOADS >= 1 ? a.resid + (long)b * a.r_bs : a.y);
    const unsigned ylane = 4u * ((unsigned)rr * (unsigned)a.y_ld + 4u * (unsigned)pc);
    const unsigned rlane = 4u * ((unsigned)rr * (unsigned)a.r_ld + 4u * (unsigned)pc);
    auto yterm = [&](int n, int i) { return 4u * ((unsigned)(row0 + 8 * i) * (unsigned)a.y_ld + (unsigned)(col0 + 32 * n)); };
    auto rterm = [&](int n, int i) { return 4u * ((unsigned)(row0 + 8 * i) * (unsigned)a.r_ld + (unsigned)(col0 + 32 * n)); };
    auto ld4 = [&](buf_rsrc rs, unsigned lane_off, unsigned uni) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, uni, AUXV ? AUXV : KX_EPI_LD_AUX));
    };
    float bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = a.bias ? a.bias[row0 + 8 * i + rr] : 0.f;
    const float div_d = a.out_div, div_rd = 1.0f / a.out_div;
    constexpr int UT = LOADS == 2 ? 1 : 2;  // tiles per load unit
    constexpr int NU = GT / UT;
    f32x4 rv[2][UT][4], yv[2][LOADS == 2 ? UT : 1][4];
    auto load_unit = [&](int u, f32x4 (&rvu)[UT][4], f32x4 (&yvu)[LOADS == 2 ? UT : 1][4]) {
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rvu[t][i] = ld4(rbuf, rlane, rterm(u * UT + t, i));
                if (LOADS == 2) yvu[t][i] = ld4(ybuf, ylane, yterm(u * UT + t, i));
            }
    };
    if (LOADS >= 1) load_unit(0, rv[0], yv[0]);
    float rs[4] = {0.f, 0.f, 0.f, 0.f}, rq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (LOADS >= 1 && u + 1 < NU) load_unit(u + 1, rv[(u + 1) & 1], yv[(u + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
            const int n = u * UT + t;
            scatter(n);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = *reinterpret_cast<const f32x4*>(scr + (8 * i + rr) * P + 4 * pc);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = __builtin_fmaf(v[c], acc_scale, bias4[i]);
                    if (LOADS >= 1) x += rv[u & 1][t][i][c];
                    if (LOADS == 2) x += yv[u & 1][t][i][c];
                    x *= a.out_mul;
                    if (LOADS == 2) x = div_const(x, div_d, div_rd);
                    v[c] = x;
                    rs[i] += x;
                    rq[i] = __builtin_fmaf(x, x, rq[i]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ybuf, ylane,
                                                       yterm(n, i), AUXV ? AUXV : KX_EPI_ST_AUX);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (want_stats) {
        // a row's partial lives in its 8 lanes (pieces 0..7): xor-1, xor-2 (quad_perm), then the mirror inside each half row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = rs[i], q = rq[i];
            s = dpp_add<0xB1>(s);
            q = dpp_add<0xB1>(q);
            s = dpp_add<0x4E>(s);
            q = dpp_add<0x4E>(q);
            s = dpp_add<0x141>(s);  // row_half_mirror: lane j <-> 7 - j inside each group of 8
            q = dpp_add<0x141>(q);
            if (pc == 0)
                a.stat_part[((long)b * a.Cout + row0 + 8 * i + rr) * a.stat_tiles + stat_slot] = make_float2(s, q);
        }
    }
}

template <int LOADS, int AUXV = 0>
__device__ __forceinline__ void conv_store_wide4(const ConvArgs& a, f32x16 (&acc)[1][4], float acc_scale, int b, int row0,
                                                 int col0, int lane, int stat_slot, float* scr) {
    const int r = lane & 31, h = lane >> 5;
    auto sc = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[0][n][e];
    };
    conv_store_wide4_t<LOADS, 32, decltype(sc)&, 4, AUXV>(a, sc, acc_scale, b, row0, col0, lane, stat_slot, scr);
}

// Dispatcher of the direct-A kernels for one 32-row x 128-column group: the wide form where it applies (plain channel-major
// store, the whole group inside the tensor), else the general forms below.
template <int EB, bool STREAM = false>
__device__ __forceinline__ void conv_store_group(const ConvArgs& a, f32x16 (&acc)[1][4], float acc_scale, int b, int row0,
                                                 int col0, int r, int h, int ncols, int Lout, int stat_slot, float2* stat_scr,
                                                 float* wide_scr);

// EB = rows per load batch of the read-modify-write path
template <int MT, int NT, int EB, bool GELU>
__device__ __forceinline__ void conv_store_tile(const ConvArgs& a, f32x16 (&acc)[MT][NT], float acc_scale, int b,
                                                int row0, int col0, int r, int h, int ncols, int Lout, int stat_slot,
                                                float2* stat_scr = nullptr) {
    if (a.store == ST_NORMAL || a.store == ST_TMAJOR) {
        const bool full = row0 + MT * 32 <= a.Cout && col0 + NT * 32 <= ncols;  // wave-uniform
        const bool res = a.resid != nullptr && a.store == ST_NORMAL;
        const bool accum = a.accum != 0 && a.store == ST_NORMAL;
        if (!full)
            conv_store_rmw<MT, NT, EB, 3, false, GELU>(a, acc, acc_scale, b, row0, col0, r, h, ncols, stat_slot, stat_scr);
        else if (res && accum)
            conv_store_rmw<MT, NT, EB, 2, true, GELU>(a, acc, acc_scale, b, row0, col0, r, h, ncols, stat_slot, stat_scr);
        else if (res)
            conv_store_rmw<MT, NT, EB, 1, true, GELU>(a, acc, acc_scale, b, row0, col0, r, h, ncols, stat_slot, stat_scr);
        else if (!accum)
            conv_store_rmw<MT, NT, EB, 0, true, GELU>(a, acc, acc_scale, b, row0, col0, r, h, ncols, stat_slot, stat_scr);
        else  // accumulate without a residual: not a form the graph has; the edge path handles any combination
            conv_store_rmw<MT, NT, EB, 3, false, GELU>(a, acc, acc_scale, b, row0, col0, r, h, ncols, stat_slot, stat_scr);
        return;
    }
    // ST_UPSCATTER: polyphase transposed conv, GEMM row = co * s + p (PHASE FASTEST), column q -> out[co][s*q + p - pad].
    // With the phase fastest, the s phases of an output line are rows of ONE wave, stored by back-to-back instructions, so
    // the line is whole in L2 when it leaves for HBM.  (Rows (p, co) put every phase in another row tile = another
    // workgroup: lines left L2 partly written, and the PMC counters showed 2.6x the algorithmic writes plus the
    // read-modify-write fetches, 3.1x the algorithmic traffic in all: profiles/r04_pmc_conv_traffic_f16x3.json.)
    const int rows_here = MT * 32;
    const int q_lo = col0, q_hi = col0 + NT * 32 - 1;
    // interior block: all rows valid, all columns valid, every phase of every column lands inside [0, Lout) ->
    // branch-free form with scalar row terms (row / s by a 16-bit reciprocal: exact for s <= 12 and rows < 8192)
    const bool interior = row0 + rows_here <= a.Cout && q_hi < ncols && a.up_s * q_lo - a.up_pad >= 2 &&
                          a.up_s * q_hi + (a.up_s - 1) - a.up_pad < Lout && a.up_s <= 12 && a.Cout < 8192;
    if (interior) {
        const buf_rsrc ybuf = make_buf(a.y + (long)b * a.y_bs);
        const buf_rsrc rbuf = make_buf(a.resid ? a.resid + (long)b * a.r_bs : a.y);
        const bool has_res = a.resid != nullptr;
        const unsigned inv_s = (65536u + (unsigned)a.up_s - 1u) / (unsigned)a.up_s;
        const int brow = row0 + r + 32 * h;  // bias of the wave's rows, one per lane (row -> co below)
        const float bias_lane = a.bias ? a.bias[((unsigned)brow * inv_s) >> 16] : 0.f;
        unsigned qoff[NT];  // byte offset of column q inside an output row: 4 * s * q
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) qoff[nt] = 4u * (unsigned)(a.up_s * (col0 + nt * 32 + r));
        // rows come in pairs (c, c + 4) split over the half-waves; their uniform byte terms for y and resid
        auto row_terms = [&](int d, unsigned& yt, unsigned& rt) {
            const int row = row0 + d;  // (wave-uniform)
            const int co = (int)(((unsigned)row * inv_s) >> 16), pp = row - co * a.up_s;
            const int t = pp - a.up_pad + a.up_off;
            yt = 4u * (unsigned)(co * a.y_ld + t);
            rt = 4u * (unsigned)(co * a.r_ld + t);
        };
        // residual loads of a whole 32-row tile are issued before its stores (see conv_store_rmw on vmcnt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                float rv[4][NT];
                unsigned yts[4];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = eg * 4 + e4;
                    const int c = mt * 32 + (e & 3) + 8 * (e >> 2);
                    unsigned y0, r0, y1, r1;
                    row_terms(c, y0, r0);
                    row_terms(c + 4, y1, r1);
                    yts[e4] = h ? y1 : y0;
                    const unsigned rts = h ? r1 : r0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) rv[e4][nt] = has_res ? buf_load(rbuf, qoff[nt] + rts, 0u) : 0.f;
                }
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = eg * 4 + e4;
                    const int c = mt * 32 + (e & 3) + 8 * (e >> 2);
                    const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bias_lane), c));
                    const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bias_lane), c + 4));
                    const float bvv = h ? b1 : b0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        buf_store(ybuf, qoff[nt] + yts[e4], 0u, __builtin_fmaf(acc[mt][nt][e], acc_scale, bvv) + rv[e4][nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    float* yb = a.y + (long)b * a.y_bs;
    const float* rb = a.resid ? a.resid + (long)b * a.r_bs : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = row0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row >= a.Cout) continue;
            const int co = row / a.up_s;
            const int p = row - co * a.up_s;
            const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = col0 + nt * 32 + r;
                if (col >= ncols) continue;
                const int tout = a.up_s * col + p - a.up_pad;
                if (tout < 0 || tout >= Lout) continue;
                const float v = __builtin_fmaf(acc[mt][nt][e], acc_scale, bv);
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

#ifndef KX_EPI_WIDE
#define KX_EPI_WIDE 1
#endif
// STREAM: the kernel has the non-temporal variant of the wide store (the tile forms of chip-filling grids only: a launch on the
// small-grid forms never has an output large enough to be streamed, and their 168-register budget has no room for a second copy)
template <int EB, bool STREAM>
__device__ __forceinline__ void conv_store_group(const ConvArgs& a, f32x16 (&acc)[1][4], float acc_scale, int b, int row0,
                                                 int col0, int r, int h, int ncols, int Lout, int stat_slot, float2* stat_scr,
                                                 float* wide_scr) {
    const bool full = row0 + 32 <= a.Cout && col0 + 128 <= ncols;  // wave-uniform
    if (KX_EPI_WIDE && a.store == ST_NORMAL && full && a.merge_T == 0 && !(a.dbg & 16384)) {
        const int lane = r + 32 * h;
        const bool res = a.resid != nullptr, accum = a.accum != 0;
        if constexpr (STREAM) {
            if (a.epi_stream) {  // (see conv_store_group16)
                if (res && accum) {
                    conv_store_wide4<2, 2>(a, acc, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
                    return;
                }
                if (res && !accum) {
                    conv_store_wide4<1, 2>(a, acc, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
                    return;
                }
                if (!res && !accum) {
                    conv_store_wide4<0, 2>(a, acc, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
                    return;
                }
            }
        }
        if (res && accum) {
            conv_store_wide4<2>(a, acc, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
            return;
        }
        if (res && !accum) {
            conv_store_wide4<1>(a, acc, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
            return;
        }
        if (!res && !accum) {
            conv_store_wide4<0>(a, acc, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
            return;
        }
    }
    conv_store_tile<1, 4, EB, false>(a, acc, acc_scale, b, row0, col0, r, h, ncols, Lout, stat_slot, stat_scr);
}

// ---- the same for accumulators of v_mfma_f32_16x16x32_f16 (conv1d_f16x3_da_kernel, S16 form): the wave's 32 rows x 128 columns
// are 2 x 8 blocks of 16 x 16, lane l holding rows 4 (l >> 4) + q, column l & 15 of a block.  The wide form scatters them into the
// same LDS transpose (pitch 36: the four row groups of one write land on different banks) and is otherwise the code above; the
// general forms get the group re-laid into the 32 x 32 C/D layout through the same scratch first.
using f32x4v = __attribute__((ext_vector_type(4))) float;
template <int EB, int NBT, int GT, bool STREAM = false>
__device__ __forceinline__ void conv_store_group16(const ConvArgs& a, f32x4v (&acc)[2][NBT], const int blk0, float acc_scale, int b,
                                                   int row0, int col0, int lane, int ncols, int Lout, int stat_slot,
                                                   float2* stat_scr, float* wide_scr) {
    const bool full = row0 + 32 <= a.Cout && col0 + 32 * GT <= ncols;  // wave-uniform
    const int c16 = lane & 15, rg = lane >> 4;
    auto scatter = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int q = 0; q < 4; ++q) wide_scr[(16 * i + 4 * rg + q) * 36 + 16 * jb + c16] = acc[i][blk0 + 2 * n + jb][q];
    };
    if (KX_EPI_WIDE && a.store == ST_NORMAL && full && a.merge_T == 0 && !(a.dbg & 16384)) {
        const bool res = a.resid != nullptr, accum = a.accum != 0;
        // epi_stream (launch-uniform): the output tensor is far larger than L2 and MALL -- the next layer reads it back from HBM
        // whatever happens -- so its stores and the residual loads are non-temporal (measured at batch 64: -0.4 ms per step on the
        // F8 forms, -0.3 more on the 3-tap forms; WITHOUT the size rule, on every direct-A launch alike: +5 ms -- the token-axis and
        // decoder tensors do fit the caches --; at batch 1 +0.1 ms: profiles/r05_f16f8_form.txt)
        if constexpr (STREAM) {
            if (a.epi_stream) {
                if (res && accum) {
                    conv_store_wide4_t<2, 36, decltype(scatter)&, GT, 2>(a, scatter, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
                    return;
                }
                if (res && !accum) {
                    conv_store_wide4_t<1, 36, decltype(scatter)&, GT, 2>(a, scatter, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
                    return;
                }
                if (!res && !accum) {
                    conv_store_wide4_t<0, 36, decltype(scatter)&, GT, 2>(a, scatter, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
                    return;
                }
            }
        }
        if (res && accum) {
            conv_store_wide4_t<2, 36, decltype(scatter)&, GT>(a, scatter, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
            return;
        }
        if (res && !accum) {
            conv_store_wide4_t<1, 36, decltype(scatter)&, GT>(a, scatter, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
            return;
        }
        if (!res && !accum) {
            conv_store_wide4_t<0, 36, decltype(scatter)&, GT>(a, scatter, acc_scale, b, row0, col0, lane, stat_slot, wide_scr);
            return;
        }
    }
    const int r = lane & 31, h = lane >> 5;
    f32x16 g[1][GT];
#pragma unroll
    for (int n = 0; n < GT; ++n) {
        scatter(n);
#pragma unroll
        for (int e = 0; e < 16; ++e) g[0][n][e] = wide_scr[((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the scratch is the statistics scratch of the general form)
    conv_store_tile<1, GT, EB, false>(a, g, acc_scale, b, row0, col0, r, h, ncols, Lout, stat_slot, stat_scr);
}

}  // namespace kx
