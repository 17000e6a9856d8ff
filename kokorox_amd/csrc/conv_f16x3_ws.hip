// Wave-specialised, persistent form of the f16x3 conv (conv_f16x3.hip holds the arithmetic and the operand images;
// this file only changes WHO does what inside a workgroup, and when).
//
// Measured on the one-role kernel (profiles/r01_f16x3_workgroup_stamps.txt, r01_pmc_sq_conv_f16x3.txt): its vector
// half (loads, AdaIN + snake, f16 split, LDS writes, LDS-DMA issue) and its matrix half run one after the other in
// every wave, and the co-resident workgroup only fills the gaps by chance (VALU/MFMA co-execution 12 %).  Here the
// two halves are different waves of one 8-wave workgroup:
//   waves 0-7  consumers (two per SIMD): ds_read_b128 fragments + v_mfma_f32_32x32x16_f16, nothing else in the main
//              loop; each owns a 64 x 64 (or 32 x 64) block of accumulators and runs the shared epilogue
//              (conv_epilogue.h).
//   waves 8-11 producers: stage input chunk c+1 (coalesced buffer loads -> AdaIN affine -> snake / leaky -> hi/lo
//              split -> [time][8 ch] image in LDS), while the consumers multiply chunk c.
//              The producers also copy the weight pieces with LDS-DMA, two steps ahead (three piece buffers): under
//              this kernel's load a weight load takes ~2000 cycles from L2, about one step of matrix work.
// Each SIMD holds two consumer waves and one producer wave (3 x 168 registers), so vector and matrix instructions
// come from different waves and co-issue.  One s_barrier per step (= one weight piece of up to TKW taps); X is double-buffered.
// The workgroup is persistent: it walks tiles  blockIdx.x, + gridDim.x, ...  and the producers start the next
// tile's first chunk while the consumers store the current one, so neither the prologue nor (on the memory side)
// the epilogue of a tile leaves the CU idle although only one workgroup fits a CU.
//
// Eligible launches (conv16_use_ws): 128-row weight tiles, stride 1, taps spanning <= 64 columns, not the merged
// token-axis GEMM form.  Everything else stays on conv1d_f16x3_kernel.
#include "conv_f16x3_common.h"

namespace kx {

constexpr int WS_TKW = 3;      // most taps per weight piece (24 KiB per LDS buffer at 128 rows)
constexpr int WS_NWB = 3;      // weight piece buffers (LDS-DMA copies run two steps ahead)
constexpr int WS_HALO = 64;    // window = BN + 64 columns: every shape of the graph has (K-1)*dil <= 50

// Opt-in (KX_WS=1, or conv mode 2 of the test hook): correct on every parity test, but measured no faster than the
// one-role kernel on this graph (146 vs 138 ms per step) - DESIGN.md §3 has the measurements and what bounds both.
bool conv16_ws_eligible(int BM, int K, int dil, int stride, int merged) {
    return BM == 128 && stride == 1 && !merged && K >= 2 && (K - 1) * dil <= WS_HALO;
}
bool conv16_use_ws(int BM, int K, int dil, int stride, int merged) {
    static const int on = getenv("KX_WS") ? atoi(getenv("KX_WS")) : 0;
    return on && conv16_ws_eligible(BM, K, dil, stride, merged);
}

// End of a producer step: wait until all but the (EXTRA + ng) youngest vector-memory operations of this wave have
// completed and every LDS write has been performed, then join the step's barrier.  The youngest operations are, in
// issue order, the EXTRA loads of load_raw() and the ng LDS-DMA instructions of the piece two steps ahead; everything
// older - in particular the piece the consumers read NEXT step - has landed (vector memory operations of a wave
// complete in order).
template <int EXTRA>
__device__ __forceinline__ void ws_producer_barrier(int ng, bool skip = false) {
    if (skip) {  // timing ablation (KX_DBG bit 16): no waits, no barriers
        return;
    }
    switch (ng) {
        case 0: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(EXTRA + 0) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(EXTRA + 2) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(EXTRA + 4) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(EXTRA + 6) : "memory"); break;
    }
}

template <int BN, int WM, int WN, int ACT>
__global__ __launch_bounds__((WM * WN + 4) * 64, (WM * WN + 4) / 4)
void conv1d_f16x3_ws_kernel(const ConvArgs a) {
    constexpr int BM = 128;
    constexpr int MT = BM / WM / 32;
    constexpr int NT = BN / WN / 32;
    constexpr int NCW = WM * WN;  // consumer waves (waves NCW .. NCW + 3 are the producers)
    constexpr int NWV = NCW + 4;  // all waves
    static_assert((NCW == 4 || NCW == 8) && MT >= 1 && NT >= 1, "4 or 8 consumer waves");
    constexpr int XWP = BN + WS_HALO;   // window pitch (columns)
    constexpr int NB = XWP / 64;        // 64-column blocks of the window
    constexpr int tap_units = 4 * BM;   // uint4 per tap: [hi|lo][h][BM]
    constexpr int piece_units = WS_TKW * tap_units;
    constexpr int xbuf_units = 4 * XWP;  // [hi|lo][octet][XWP]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ws[];
    uint4* Wbuf = reinterpret_cast<uint4*>(smem_ws);
    uint4* Xbuf = Wbuf + WS_NWB * piece_units;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, dil = a.dil;
    const int n_chunks = a.n_chunks16;
    const int n_pieces = (K + WS_TKW - 1) / WS_TKW;
    const int tpp = (K + n_pieces - 1) / n_pieces;  // taps per piece, as even as possible
    const int n_steps = n_chunks * n_pieces;
    const int ntx = a.ws_ntx, nty = a.ws_nty, n_tiles = a.ws_tiles;

    // ---- the workgroup's walk over tiles: blockIdx.x, + gridDim.x, ..., skipping tiles past an utterance's end ------
    auto tile_cols = [&](int t, int& b, int& ct, int& tx, int& Lin, int& Lout) -> int {
        tx = t % ntx;
        const int ty = t / ntx;
        ct = ty % nty;
        b = __builtin_amdgcn_readfirstlane(ty / nty);  // (wave-uniform: lets the length lookups be scalar loads, which
                                                        // do not touch the vector-memory counter of the weight pipeline)
        Lin = a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
        Lout = a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
        return (a.store == ST_UPSCATTER) ? (Lin + 1) : Lout;
    };
    auto tile_valid = [&](int t) -> bool {
        int b, ct, tx, Lin, Lout;
        return tx = 0, tile_cols(t, b, ct, tx, Lin, Lout) > tx * BN;
    };
    auto next_tile = [&](int t) -> int {  // (>= n_tiles: the walk has ended)
        do t += gridDim.x;
        while (t < n_tiles && !tile_valid(t));
        return t;
    };
    int first_tile = blockIdx.x;
    if (first_tile < n_tiles && !tile_valid(first_tile)) first_tile = next_tile(first_tile);
    if (first_tile >= n_tiles) return;

    int g = 0;  // global step (counted across the tiles of the walk): piece g lives in W buffer g % WS_NWB

    if (wave >= NCW) {
        // =================================== producers ===================================================
        const int pw = wave - NCW;          // channel quad of the 16-channel chunk this wave stages
        const int og = pw >> 1;             // octet of that quad
        const bool has_norm = a.nmean != nullptr;
        const int up2 = a.in_up2;
        const int cmax_in = a.Cin - 1;
        const unsigned row_bytes = 4u * (unsigned)a.x_ld;
        for (int tile = first_tile; tile < n_tiles; tile = next_tile(tile)) {
            int b, ct, tx, Lin, Lout;
            (void)tile_cols(tile, b, ct, tx, Lin, Lout);
            const int t0 = tx * BN;
            const float* xb = a.x + (long)b * a.x_bs;
            const buf_rsrc xrs = make_buf(xb);
            const int p0 = t0 - a.pad;
            const int Lsrc = up2 ? ((Lin + 1) >> 1) : Lin;

            // per-lane byte offsets of this lane's NB window columns and their validity (zero padding)
            unsigned xoff[NB];
            float keep[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int p = p0 + lane + 64 * j;
                keep[j] = (p >= 0 && p < Lin) ? a.x_prescale : 0.f;  // (zero padding and the activation pre-scale)
                int pi = up2 ? (p >> 1) : p;
                pi = pi < 0 ? 0 : (pi >= Lsrc ? Lsrc - 1 : pi);
                xoff[j] = 4u * (unsigned)pi;
            }
            float raw[NB][4];
            // Per-channel parameters of the pending chunk travel in four registers: lane c (mod 4) holds mean /
            // scale / shift / alpha of the wave's channel c.  They are loaded together with the raw values (one chunk
            // ahead) and handed out with v_readlane when the chunk is transformed.  Without a norm the three loads
            // still happen (from the first floats of the input, values unused): one code path.
            const float* pm_src = has_norm ? a.nmean + (long)b * a.n_bs : xb;
            const float* ps_src = has_norm ? a.nscale + (long)b * a.n_bs : xb;
            const float* ph_src = has_norm ? a.nshift + (long)b * a.n_bs : xb;
            float pm = 0.f, ps = 1.f, ph = 0.f, pa = 1.f;
            auto load_raw = [&](int ch) {
                {
                    const int ci = ch * CK16 + pw * 4 + (lane & 3);
                    const int cc = ci < cmax_in ? ci : cmax_in;
                    pm = pm_src[cc];
                    ps = ps_src[cc];
                    ph = ph_src[cc];
                    if (ACT == ACT_SNAKE) pa = a.alpha[cc];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ci = ch * CK16 + pw * 4 + c;
                    const unsigned rt = (unsigned)(ci < cmax_in ? ci : cmax_in) * row_bytes;  // (wave-uniform)
#pragma unroll
                    for (int j = 0; j < NB; ++j) raw[j][c] = buf_load(xrs, xoff[j], rt);
                }
            };
            auto lane_bcast = [](float v, int l) {
                return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
            };
            // transform the staged raw values of chunk ch and write them into X buffer (ch & 1)
            auto stage = [&](int ch) {
                if (a.dbg & 1) {  // timing ablation: no transform, but the loads are still waited for
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(raw[j][c]));
                    asm volatile("" ::"v"(pm), "v"(ps), "v"(ph), "v"(pa));
                    return;
                }
                float m[4], sc[4], hh[4], al[4], ial[4], cv[4];
                const float pia = 1.0f / pa;  // (one division per chunk and lane instead of one per channel)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ci = ch * CK16 + pw * 4 + c;
                    m[c] = has_norm ? lane_bcast(pm, c) : 0.f;
                    sc[c] = has_norm ? lane_bcast(ps, c) : 1.f;
                    hh[c] = has_norm ? lane_bcast(ph, c) : 0.f;
                    al[c] = (ACT == ACT_SNAKE) ? lane_bcast(pa, c) : 1.f;
                    ial[c] = (ACT == ACT_SNAKE) ? lane_bcast(pia, c) : 1.f;
                    cv[c] = (ci <= cmax_in) ? 1.f : 0.f;  // channels past Cin (ragged last chunk) are zeros
                }
                unsigned char* xw = reinterpret_cast<unsigned char*>(Xbuf + (ch & 1) * xbuf_units) + (pw & 1) * 8;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    float y[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = in_act<ACT>(__builtin_fmaf(raw[j][c] - m[c], sc[c], hh[c]), a.slope, al[c], ial[c]);
                        // zero padding comes after the activation (a multiply keeps the activation branch-free;
                        // masked positions hold clamped-address tensor values, i.e. finite numbers)
                        y[c] = v * (keep[j] * cv[c]);
                    }
                    unsigned h0, l0, h1, l1;
                    split_pair(y[0], y[1], h0, l0);
                    split_pair(y[2], y[3], h1, l1);
                    const int u = lane + 64 * j;
                    *reinterpret_cast<uint2*>(xw + ((0 * 2 + og) * XWP + u) * 16) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(xw + ((1 * 2 + og) * XWP + u) * 16) = make_uint2(l0, l1);
                }
            };

            // prologue of the tile (runs beside the consumers' epilogue of the previous tile): chunk 0 into LDS, the
            // raw values of chunk 1 in flight
            constexpr int NRAW = NB * 4 + 3 + (ACT == ACT_SNAKE ? 1 : 0);  // loads of one load_raw()
            // LDS-DMA copy of the weight piece of the tile's step s into W buffer (g0 + s) % 3.  A piece is taps * 8
            // segments of 1 KiB; producer wave pw copies the 2 * taps consecutive segments pw * 2 * taps ...  M0 (the
            // LDS base of an LDS-DMA instruction) is written ONCE per wave and piece and the segments are addressed
            // through the instruction's immediate offset, which moves the global and the LDS address together: with
            // one s_mov m0 per copy instruction each copy took ~350 cycles to issue (the scalar write of M0 waits
            // until the previous LDS-DMA instruction has consumed it) and the producers were the last wave at every
            // barrier.  Returns the number of copy instructions issued by this wave.
            const int g0 = g;
            const bool nobar = (a.dbg & 16) != 0;
            const uint4* wbase = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_chunks * K * tap_units;
            auto issue_piece = [&](int s) -> int {
                if (s >= n_steps || (a.dbg & 2)) return 0;
                const int ch = s / n_pieces, pc = s - ch * n_pieces;
                const int tp = pc * tpp;
                const int taps = (K - tp) < tpp ? (K - tp) : tpp;
                const int first = pw * 2 * taps;  // first segment of this wave
                const uint4* src = wbase + ((long)ch * K + tp) * tap_units + first * 64 + lane;
                uint4* dst = Wbuf + ((g0 + s) % WS_NWB) * piece_units + first * 64;
                const unsigned lds_addr = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) void*)dst);
                // (immediate offsets reach 4095 bytes: segments 0..3 from the first base, 4..5 from a second one)
                if (taps == 3) {
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %1, off\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                                 "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %3, off\n\t"
                                 "global_load_lds_dwordx4 %3, off offset:1024"
                                 ::"s"(lds_addr), "v"(src), "s"(lds_addr + 4096u), "v"(src + 256) : "memory");
                } else if (taps == 2) {
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %1, off\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:3072"
                                 ::"s"(lds_addr), "v"(src) : "memory");
                } else {
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %1, off\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:1024"
                                 ::"s"(lds_addr), "v"(src) : "memory");
                }
                return 2 * taps;
            };
            // prologue of the tile (runs beside the consumers' epilogue of the previous tile): pieces 0 and 1, chunk 0,
            // then the raw values of chunk 1 go in flight
            issue_piece(0);
            issue_piece(1);
            load_raw(0);
            stage(0);
            if (n_chunks > 1) {
                load_raw(1);
                ws_producer_barrier<NRAW>(0, nobar);
            } else {
                ws_producer_barrier<0>(0, nobar);
            }
            int s = 0;
            for (int ch = 0; ch < n_chunks; ++ch) {
                if (ch + 1 < n_chunks) {
                    stage(ch + 1);  // (the compiler waits for the raw loads here)
                    if (ch + 2 < n_chunks) {
                        load_raw(ch + 2);
                        asm volatile("" ::: "memory");  // keep the copies below behind the loads (vmcnt order)
                        ws_producer_barrier<NRAW>(issue_piece(s + 2), nobar);
                    } else {
                        ws_producer_barrier<0>(issue_piece(s + 2), nobar);
                    }
                } else {
                    ws_producer_barrier<0>(issue_piece(s + 2), nobar);
                }
                ++s;
                for (int pc = 1; pc < n_pieces; ++pc, ++s) ws_producer_barrier<0>(issue_piece(s + 2), nobar);
            }
            g += n_steps;
        }
        return;
    }

    // ======================================= consumers =======================================================
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    for (int tile = first_tile; tile < n_tiles; tile = next_tile(tile)) {
        int b, ct, tx, Lin, Lout;
        const int ncols = tile_cols(tile, b, ct, tx, Lin, Lout);
        const int t0 = tx * BN;
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // diagnostic stamps (KX_STAMP, consumer wave 0, the workgroup's SECOND tile = steady state): real time
        // (100 MHz) and shader cycles at the tile's phases, plus the cycles spent waiting at the step barriers
        const bool stamp = a.stamps != nullptr && wave == 0 && tile == (int)(blockIdx.x + gridDim.x);
        unsigned long long sr0 = 0, sc0 = 0, sr1 = 0, sc1 = 0, sr2 = 0, sc2 = 0, bar_cyc = 0;
        if (stamp) {
            sr0 = __builtin_amdgcn_s_memrealtime();
            sc0 = __builtin_readcyclecounter();
        }
        __builtin_amdgcn_s_setprio(2);
        // chunk 0 is in LDS; so is the tile's first weight piece (written in the last step of the previous tile, or
        // - first tile - just above: this wave's part of it must have been performed before the barrier)
        if (!(a.dbg & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (stamp) {
            sr1 = __builtin_amdgcn_s_memrealtime();
            sc1 = __builtin_readcyclecounter();
        }
        for (int ch = 0; ch < n_chunks; ++ch)
        for (int pc = 0; pc < n_pieces; ++pc, ++g) {
            const int tp = pc * tpp;
            const int taps = (K - tp) < tpp ? (K - tp) : tpp;
            const uint4* wl = Wbuf + (g % WS_NWB) * piece_units + h * BM + wm * (MT * 32) + r;
            const uint4* xl = Xbuf + (ch & 1) * xbuf_units + h * XWP + wn * (NT * 32) + r;
            // P = {a_lo, b_hi} feeds the first MT*NT MFMAs of a tap, Q = {a_hi, b_lo} (+ b_hi again) the other
            // 2*MT*NT: Q(t) is read under the P-phase of tap t, P(t+1) under its Q-phase
            auto load_P = [&](int tt, half8 (&al)[MT], half8 (&bh)[NT]) {
                const uint4* wt = wl + tt * tap_units;
                const uint4* xt = xl + (tp + tt) * dil;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) al[mt] = *reinterpret_cast<const half8*>(&wt[2 * BM + mt * 32]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(&xt[nt * 32]);
            };
            auto load_Q = [&](int tt, half8 (&ah)[MT], half8 (&bl)[NT]) {
                const uint4* wt = wl + tt * tap_units;
                const uint4* xt = xl + (tp + tt) * dil;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ah[mt] = *reinterpret_cast<const half8*>(&wt[mt * 32]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bl[nt] = *reinterpret_cast<const half8*>(&xt[2 * XWP + nt * 32]);
            };
            if (!(a.dbg & 4)) {
                half8 al_[2][MT], bh[2][NT], ah[MT], bl[NT];
                load_P(0, al_[0], bh[0]);
#pragma unroll
                for (int tt = 0; tt < WS_TKW; ++tt) {
                    if (tt < taps) {
                        load_Q(tt, ah, bl);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_[tt & 1][mt], bh[tt & 1][nt], acc[mt][nt], 0, 0, 0);
                        if (tt + 1 < WS_TKW && tt + 1 < taps) load_P(tt + 1, al_[(tt + 1) & 1], bh[(tt + 1) & 1]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[tt & 1][nt], acc[mt][nt], 0, 0, 0);
                            }
                    }
                }
            }
            // all fragment reads of this step have returned (the MFMAs consumed them) and this wave's weight segments of
            // the next step are written: the producers may overwrite X, everyone may read the next piece
            if (stamp) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long tb = __builtin_readcyclecounter();
                asm volatile("s_barrier" ::: "memory");
                bar_cyc += __builtin_readcyclecounter() - tb;
            } else if (!(a.dbg & 16)) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (stamp) {
            sr2 = __builtin_amdgcn_s_memrealtime();
            sc2 = __builtin_readcyclecounter();
        }
        if (a.dbg & 8) continue;
        // (the producers are already staging the next tile: its loads run beside these stores)
        conv_store_tile<MT, NT, (NCW == 8 ? 2 : EPI_ROWS), false>(a, acc, a.w_unscale, b, ct * BM + wm * (MT * 32), t0 + wn * (NT * 32), r, h,
                                                 ncols, Lout, tx * WN + wn);
        if (stamp && lane == 0) {  // stamps leave through a buffer of their own that nothing else reads
            unsigned long long* o = a.stamps + (unsigned long long)blockIdx.x * 8;
            __builtin_amdgcn_s_waitcnt(0);
            o[0] = sr0;                                        // tile start (before the first barrier)
            o[1] = sr1;                                        // first barrier passed: main loop begins
            o[2] = sr2;                                        // main loop done
            o[3] = __builtin_amdgcn_s_memrealtime();           // epilogue done
            o[4] = sc2 - sc1;                                  // shader cycles of the main loop
            o[5] = __builtin_readcyclecounter() - sc2;         // shader cycles of the epilogue
            o[6] = bar_cyc;                                    // cycles inside the step barriers
            o[7] = (unsigned long long)n_steps | ((sc1 - sc0) << 32);  // steps | cycles waiting for the prologue
        }
    }
}

static int device_cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int BN, int WM, int WN, int ACT>
static void launch_ws_inst(ConvArgs a, int B, int max_cols, hipStream_t s) {
    auto kern = conv1d_f16x3_ws_kernel<BN, WM, WN, ACT>;
    constexpr size_t lds = 16 * ((size_t)WS_NWB * WS_TKW * 4 * 128 + (size_t)2 * 4 * (BN + WS_HALO));
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        KX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    a.ws_ntx = (max_cols + BN - 1) / BN;
    a.ws_nty = (a.Cout + 127) / 128;
    const long tiles = (long)a.ws_ntx * a.ws_nty * B;
    KX_REQUIRE(a.ws_ntx > 0 && a.ws_nty > 0 && B > 0 && tiles < (1L << 30), "conv1d f16x3 ws: bad tile count");
    a.ws_tiles = (int)tiles;
    // one persistent workgroup per CU (256 registers per wave: a second one would not be resident anyway)
    const int cus = device_cu_count();
    static const int per_cu = getenv("KX_WS_WG_PER_CU") ? atoi(getenv("KX_WS_WG_PER_CU")) : 1;
    const long want = (long)cus * (per_cu > 0 ? per_cu : 1);
    const int grid = (int)(tiles < want ? tiles : want);
    hipLaunchKernelGGL(kern, dim3(grid), dim3((WM * WN + 4) * 64), lds, s, a);
    KX_HIP(hipGetLastError());
}

template <int BN, int WM, int WN>
static void launch_ws_act(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    if (a.act == ACT_SNAKE)
        launch_ws_inst<BN, WM, WN, ACT_SNAKE>(a, B, max_cols, s);
    else if (a.act == ACT_LEAKY)
        launch_ws_inst<BN, WM, WN, ACT_LEAKY>(a, B, max_cols, s);
    else
        launch_ws_inst<BN, WM, WN, ACT_NONE>(a, B, max_cols, s);
}

// Tile of the wave-specialised form: 128 x 256 with the 8 consumers 2 x 4 (64 x 64 per wave).  Small grids (fewer
// 256-column tiles than half the CUs: batch 1) take 128 x 128 with the consumers 4 x 2 (32 x 64 per wave): twice
// the workgroups, and each wave still spans 64 columns, so the fused InstanceNorm partial sums cover the same
// column groups in the same order and results stay bit-identical across batch sizes.
static int ws_consumers() {  // KX_WS_CONS=8: two consumer waves per SIMD (168 registers per wave); default 4 (256)
    static const int v = getenv("KX_WS_CONS") ? atoi(getenv("KX_WS_CONS")) : 4;
    return v == 8 ? 8 : 4;
}
void conv16_ws_tile(int max_cols, int B, int Cout, int* bn, int* wn) {
    static const int force = getenv("KX_WS_BN") ? atoi(getenv("KX_WS_BN")) : 0;
    const long tiles256 = (long)((max_cols + 255) / 256) * ((Cout + 127) / 128) * B;
    const bool small = force == 128 || (force != 256 && tiles256 * 2 <= device_cu_count());
    *bn = small ? 128 : 256;
    *wn = ws_consumers() == 8 ? (small ? 2 : 4) : (small ? 1 : 2);
}

void launch_conv1d_f16x3_ws(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    KX_REQUIRE(conv16_ws_eligible(128, a.K, a.dil, a.stride, a.merge_T > 0), "conv1d f16x3 ws: launch not eligible");
    KX_REQUIRE(a.n_chunks16 == (a.Cin + CK16 - 1) / CK16 && a.w16 != nullptr, "conv1d f16x3 ws: weights not packed");
    KX_REQUIRE(a.epi != EPI_GELU_NEW, "conv1d f16x3 ws: no gelu epilogue");
    if (max_cols <= 0) return;
    int bn, wn;
    conv16_ws_tile(max_cols, B, a.Cout, &bn, &wn);
    if (ws_consumers() == 8) {
        if (bn == 128)
            launch_ws_act<128, 4, 2>(a, B, max_cols, s);
        else
            launch_ws_act<256, 2, 4>(a, B, max_cols, s);
    } else {
        if (bn == 128)
            launch_ws_act<128, 4, 1>(a, B, max_cols, s);
        else
            launch_ws_act<256, 2, 2>(a, B, max_cols, s);
    }
}

}  // namespace kx
