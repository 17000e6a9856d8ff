// conv1d as an implicit GEMM on the CDNA4 16-bit matrix cores with f32-class accuracy:
// the "f16x3" split.  Every f32 operand v is carried as two halves, hi = f16(v) and
// lo = f16(v - hi) (22 significant bits together), and each product is evaluated as
//      a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (3 x v_mfma_f32_32x32x16_f16, f32 accumulate)
// The dropped a_lo*b_lo term and the two roundings of the lo parts are each <= 2^-22 |a b|.
// Weights are pre-scaled per layer by an exact power of two (undone in the epilogue) so that their
// lo parts stay in the normal f16 range; activations are O(1), where the f16 subnormal quantum
// (2^-25 after the split) is at f32-epsilon level.  One K=16 step then costs 3 x 32 matrix cycles
// instead of the 8 x 64 of the f32 MFMA path (conv_mfma.hip): 5.3x less matrix-pipe time, which is
// what the graph is bound by.  Same ConvArgs, same epilogue, same tests.
//
// Tiling (wave64, 4 waves): rows = output channels (BM), columns = time (BN), K walked as
// 16-channel chunks x taps.  LDS images are fragment-shaped, 16 B per lane per read:
//   Xs[hi|lo][k-half h][time][8 ch]   B operand: lane (col r, h) reads Xs[.][h][t + tap*dil]
//   Ws[tap][hi|lo][h][row][8 ch]      A operand: lane (row r, h) reads Ws[tap][.][h][row]
// Both are conflict-free for ds_read_b128 (consecutive lanes -> consecutive 16-B slots).
// The AdaIN affine + snake/leaky transform, the f16 split and the [time][8ch] transposition all
// happen while the input chunk is staged, so HBM sees each activation element once per launch.
#include "conv_f16x3_common.h"

namespace kx {

template <int BM, int BN, int WM, int WN, int ACT, int TK, bool PF, int VT>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 4 : ((BM / WM / 32) * (BN / WN / 32) > 6 ? 2 : 3))
void conv1d_f16x3_kernel(const ConvArgs a) {
    constexpr int MT = BM / WM / 32;
    constexpr int NT = BN / WN / 32;
    constexpr int NWV = WM * WN;  // waves per workgroup: 4, or 8 (same tile, half the registers per wave)
    static_assert((NWV == 4 || NWV == 8) && MT >= 1 && NT >= 1, "4 or 8 waves per workgroup");
    // VT > 1 ("virtual taps", k = 1 GEMMs only): VT consecutive 16-channel chunks are staged together and walked
    // as VT taps of one super-chunk, so one barrier pair feeds VT x more matrix work
    static_assert(VT == 1 || (VT <= TK && PF), "virtual taps need TK >= VT and the prefetching build");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, ct = blockIdx.y;
    // XCD-aware column-tile order.  Workgroups are dealt round-robin over the 8 XCDs (linear block id mod 8), each with its
    // own L2, and a tile's input window overlaps its neighbour's by up to 128 columns.  With tile = blockIdx.x those
    // neighbours sit on different XCDs and the overlap is fetched from the fabric twice (PMC: 1.8x the algorithmic
    // bytes, profiles/r02_pmc_conv_traffic_f16x3.json; the k = 3 convs run at the HBM roofline because of it).  Here
    // the blocks of one XCD get a contiguous range of column tiles, so the overlap is an L2 hit.  (Speed only: the map
    // is a bijection of [0, gridDim.x) for any placement.)
    int tile_x = blockIdx.x;
    if (a.xcd_swizzle && gridDim.x >= 16) {
        const int nx = gridDim.x;
        const int off = (int)(((long)nx * (blockIdx.y + (long)gridDim.y * blockIdx.z)) & 7);  // XCD class of x = 0 in this row
        const int cls = (blockIdx.x + off) & 7;
        int start = 0;
        for (int c = 0; c < cls; ++c) {
            const int first = (c - off) & 7;  // smallest x of class c
            start += first < nx ? (nx - first + 7) >> 3 : 0;
        }
        tile_x = start + (blockIdx.x >> 3);   // x = first_cls + 8 j is the j-th block of its class: j = x >> 3 (first < 8)
    }
    const int t0 = tile_x * BN;
    const bool merged = a.merge_T > 0;

    const int Lin = merged ? a.merge_B * a.merge_T : a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = merged ? Lin : a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    const int ncols = (a.store == ST_UPSCATTER) ? (Lin + 1) : Lout;
    if (t0 >= ncols) return;

    // The prefetching build (PF) is the stride-1 form with a fixed window pitch: every B-fragment address is then
    // one per-lane base plus compile-time offsets (ds_read_b128 immediates) instead of a register each.
    constexpr int XWP_FIXED = (VT > 1) ? 128 : BN + 128;
    const int K = a.K, dil = a.dil, stride = PF ? 1 : a.stride;
    const int XW = (BN - 1) * stride + (K - 1) * dil + 1;
    const int XWp = PF ? XWP_FIXED : ((XW + 3) & ~3);
    constexpr int tap_units = 4 * BM;  // uint4 per tap: [hi|lo][h][BM]
    constexpr int piece_units = TK * tap_units;
    // LDS carve (16-B units): two weight-piece buffers, then Xs [hi|lo][octet][XWp]
    uint4* Wbuf = reinterpret_cast<uint4*>(smem16);
    uint4* Xs = Wbuf + 2 * piece_units;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int n_chunks16 = a.n_chunks16;
    const int n_chunks = (VT == 1) ? n_chunks16 : (n_chunks16 + VT - 1) / VT;  // (super-)chunks walked below
    const int n_pieces = (VT == 1) ? (K + TK - 1) / TK : 1;
    const uint4* wbase = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_chunks16 * K * tap_units;
    const float* xb = a.x + (long)b * a.x_bs;
    const int p0 = t0 * stride - a.pad;
    const bool has_norm = a.nmean != nullptr;
    const int up2 = a.in_up2;
    const int Lsrc = up2 ? ((Lin + 1) >> 1) : Lin;  // stored length of the input rows

    // asynchronous copy of weight piece (chunk ch, piece pc) into Wbuf[buf]: whole 1-KiB segments per wave
    auto issue_piece = [&](int ch, int pc, int buf) {
        const int tp = pc * TK;
        int taps = (K - tp) < TK ? (K - tp) : TK;
        const uint4* src = wbase + ((long)ch * K + tp) * tap_units;
        if (VT > 1) {  // k = 1: chunk16 c sits at c * tap_units, a super-chunk is VT of them back to back
            taps = (n_chunks16 - ch * VT) < VT ? (n_chunks16 - ch * VT) : VT;
            src = wbase + (long)ch * VT * tap_units;
        }
        uint4* dst = Wbuf + buf * piece_units;
        const int nseg = taps * tap_units / 64;
        if (a.dbg & 2) return;
        if (a.dbg & 256) {  // timing ablation: the same copy instructions, every one reading the first KiB of the image
            for (int sgm = wave; sgm < nseg; sgm += NWV) glds16(wbase + lane, dst + sgm * 64);
            return;
        }
        for (int sgm = wave; sgm < nseg; sgm += NWV) glds16(src + sgm * 64 + lane, dst + sgm * 64);
    };

    // ---- input chunk staging.  Wave w owns one channel octet of the 16-channel chunk (g = w & 1) and every
    // other 64-column block of the window (blocks (w >> 1), (w >> 1) + 2, ...): lanes run along time (coalesced
    // global loads), each lane gathers its column's 8 channels, applies AdaIN affine + activation, splits into
    // f16 hi / lo and writes two whole 16-byte slots of the [time][8 ch] images (conflict-free ds_write_b128).
    // The per-channel parameters travel in one register: lane c (mod 8) holds (mean, scale, shift, alpha) of
    // channel c, handed out with v_readlane when the chunk is transformed.  With PF both the raw values and the
    // parameters of the NEXT chunk are loaded before the MFMA loop of the current one (their latency hides
    // behind the matrix work) and are transformed + written after it.
    static_assert(NWV == 4, "the staging split assumes 4 waves");
    constexpr int NJ = PF ? XWP_FIXED / 128 : 1;  // column blocks per wave: the PF window pitch is 128 * NJ columns
    const int g = wave & 1, jb = wave >> 1;
    float raw[VT][NJ][8];
    float praw[VT][4];
    // per-lane element offsets of this lane's NJ window columns (they do not depend on the chunk)
    int xoff[NJ];
    unsigned okmask = 0;
    if (PF) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = p0 + lane + 64 * (jb + 2 * j);
            if (merged) {
                const int pc = p < Lin ? p : Lin - 1;
                const int bb = pc / a.merge_T, tt = pc - bb * a.merge_T;
                const bool ok = p < Lin && tt < a.in_len.lens[bb];
                okmask |= ok ? (1u << j) : 0u;
                xoff[j] = (int)((long)bb * a.x_bs + tt);
            } else {
                okmask |= (p >= 0 && p < Lin) ? (1u << j) : 0u;
                int pi = up2 ? (p >> 1) : p;
                pi = pi < 0 ? 0 : (pi >= Lsrc ? Lsrc - 1 : pi);
                xoff[j] = pi;
            }
        }
    }
    const float* xbase = merged ? a.x : xb;
    const int cmax_in = a.Cin - 1;
    auto load_params = [&](int ch, float (&pv)[4]) {
        const int c = ch * CK16 + g * 8 + (lane & 7);
        const int cc = c < cmax_in ? c : cmax_in;
        pv[0] = has_norm ? a.nmean[(long)b * a.n_bs + cc] : 0.f;
        pv[1] = has_norm ? a.nscale[(long)b * a.n_bs + cc] : 1.f;
        pv[2] = has_norm ? a.nshift[(long)b * a.n_bs + cc] : 0.f;
        pv[3] = (ACT == ACT_SNAKE) ? a.alpha[cc] : 1.f;  // (its reciprocal is taken when the chunk is transformed: a
                                                          // division here would wait for the load on the spot)
    };
    auto load_raw = [&](int sc) {
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            const int ch = sc * VT + vt;  // 16-channel chunk staged as virtual tap vt
            load_params(ch, praw[vt]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = ch * CK16 + g * 8 + c;
                const float* row = xbase + (long)(ci < cmax_in ? ci : cmax_in) * a.x_ld;
#pragma unroll
                for (int j = 0; j < NJ; ++j) raw[vt][j][c] = row[xoff[j]];
            }
        }
    };
    // per-channel parameters of one octet, unpacked into scalar registers once per chunk
    struct Oct { float m[8], s[8], h[8], al[8], ial[8]; };
    auto unpack_params = [&](const float (&pv)[4], Oct& o) {
        const float al_or_rcp = (lane & 8) ? 1.0f / pv[3] : pv[3];  // lanes 8..15: the reciprocals
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            o.m[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[0]), c));
            o.s[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[1]), c));
            o.h[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[2]), c));
            o.al[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_or_rcp), c));
            o.ial[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_or_rcp), c + 8));
        }
    };
    // transform one column's 8 channels and write its two 16-byte slots
    auto emit8 = [&](int vt, int ch, int u, const float (&x8)[8], const Oct& o, bool pok) {
        unsigned hp[4], lp[4];
        const float keep = pok ? a.x_prescale : 0.f;  // (zero padding and the activation pre-scale in one multiply)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            float y2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * c2 + q;
                const float y = in_act<ACT>(__builtin_fmaf(x8[c] - o.m[c], o.s[c], o.h[c]), a.slope, o.al[c], o.ial[c]);  // (explicit fma: see conv_epilogue.h)
                // zero padding comes after the activation (a multiply, so that the activation stays branch-free;
                // masked positions hold clamped-address tensor values, i.e. finite numbers)
                y2[q] = y * ((ch * CK16 + g * 8 + c <= cmax_in) ? keep : 0.f);
            }
            split_pair(y2[0], y2[1], hp[c2], lp[c2]);
        }
        Xs[(vt * 4 + 0 * 2 + g) * XWp + u] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        Xs[(vt * 4 + 1 * 2 + g) * XWp + u] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
    };
    auto stage_chunk = [&](int sc, bool from_raw) {
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            const int ch = sc * VT + vt;
            Oct o;
            if (PF && from_raw) {
                unpack_params(praw[vt], o);
                // (columns past the window but inside the fixed pitch get clamped-address garbage that no
                // fragment read ever touches)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    emit8(vt, ch, lane + 64 * (jb + 2 * j), raw[vt][j], o, ((okmask >> j) & 1u) != 0u);
            } else {
                float pv[4];
                load_params(ch, pv);
                unpack_params(pv, o);
                if (PF) {  // first chunk of the prefetching build: same hoisted offsets, loads issued here
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float x8[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const int ci = ch * CK16 + g * 8 + c;
                            x8[c] = xbase[(long)(ci < cmax_in ? ci : cmax_in) * a.x_ld + xoff[j]];
                        }
                        emit8(vt, ch, lane + 64 * (jb + 2 * j), x8, o, ((okmask >> j) & 1u) != 0u);
                    }
                } else {
                    for (int u = lane + 64 * jb; u < XW; u += 128) {
                        const int p = p0 + u;
                        int pi = up2 ? (p >> 1) : p;
                        pi = pi < 0 ? 0 : (pi >= Lsrc ? Lsrc - 1 : pi);
                        float x8[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const int ci = ch * CK16 + g * 8 + c;
                            x8[c] = xbase[(long)(ci < cmax_in ? ci : cmax_in) * a.x_ld + pi];
                        }
                        emit8(vt, ch, u, x8, o, p >= 0 && p < Lin);
                    }
                }
            }
        }
    };

    // (Tried and measured without effect on throughput: a one-off start stagger of the co-resident workgroups, and
    // distinct static s_setprio classes per CU handed out through a flag word.  The favoured workgroup's main loop
    // got 12 % shorter, the other one's as much longer.)
    unsigned long long st0 = 0, st1 = 0, st2 = 0, acc_stage = 0, acc_bar = 0, acc_pbar = 0;
    unsigned long long cyc0 = 0;
    if (a.stamps) {
        st0 = __builtin_amdgcn_s_memrealtime();
        cyc0 = __builtin_readcyclecounter();
    }
    issue_piece(0, 0, 0);
    if (!(a.dbg & 1)) stage_chunk(0, false);
    __syncthreads();  // Xs complete; the in-flight weight piece has landed (the barrier drains vmcnt)
    if (a.stamps) st1 = __builtin_amdgcn_s_memrealtime();
    int cur = 0;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const bool more = ch + 1 < n_chunks;
        if (PF && more && !(a.dbg & 1)) load_raw(ch + 1);
        // ---- taps in pieces of TK: prefetch the next piece while this one feeds the matrix pipe ---
        for (int pc = 0; pc < n_pieces; ++pc) {
            if (pc + 1 < n_pieces)
                issue_piece(ch, pc + 1, cur ^ 1);
            else if (more)
                issue_piece(ch + 1, 0, cur ^ 1);
            const int tp = pc * TK;
            int taps = (K - tp) < TK ? (K - tp) : TK;
            if (VT > 1) taps = (n_chunks16 - ch * VT) < VT ? (n_chunks16 - ch * VT) : VT;
            const uint4* Wp = Wbuf + cur * piece_units;
            // Fragment reads are software-pipelined inside the piece so that only the first read of a piece waits
            // on LDS with an idle matrix pipe.  P = {a_lo, b_hi} feeds the first 8 MFMAs of a tap, Q = {a_hi, b_lo}
            // (+ b_hi again) the other 16: Q(t) is read under the P-phase of tap t, P(t+1) under its Q-phase.
            const uint4* wl = Wp + h * BM + wm * (MT * 32) + r;
            const uint4* xl = Xs + h * XWp + (wn * (NT * 32) + r) * stride;
            auto load_P = [&](int tt, half8 (&al)[MT], half8 (&bh)[NT]) {
                const uint4* wt = wl + tt * tap_units;
                const uint4* xt = (VT > 1) ? xl + tt * 4 * XWp : xl + (tp + tt) * dil;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) al[mt] = *reinterpret_cast<const half8*>(&wt[2 * BM + mt * 32]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(&xt[nt * 32 * stride]);
            };
            auto load_Q = [&](int tt, half8 (&ah)[MT], half8 (&bl)[NT]) {
                const uint4* wt = wl + tt * tap_units;
                const uint4* xt = (VT > 1) ? xl + tt * 4 * XWp : xl + (tp + tt) * dil;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ah[mt] = *reinterpret_cast<const half8*>(&wt[mt * 32]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bl[nt] = *reinterpret_cast<const half8*>(&xt[2 * XWp + nt * 32 * stride]);
            };
            if (!(a.dbg & 4)) {
                half8 al_[2][MT], bh[2][NT], ah[MT], bl[NT];
                load_P(0, al_[0], bh[0]);
#pragma unroll
                for (int tt = 0; tt < TK; ++tt) {
                    if (tt < taps) {
                        load_Q(tt, ah, bl);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_[tt & 1][mt], bh[tt & 1][nt], acc[mt][nt], 0, 0, 0);
                        if (tt + 1 < TK && tt + 1 < taps) load_P(tt + 1, al_[(tt + 1) & 1], bh[(tt + 1) & 1]);
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
            // Everyone must be done with Wbuf[cur] (and with Xs after the last piece).  After the last piece of a
            // chunk a raw barrier with lgkmcnt(0) only is enough: the transform of the next chunk follows, and the
            // weight piece that was just issued stays in flight across it (a __syncthreads() would drain vmcnt and
            // expose the copy latency right behind a short MFMA burst); the full barrier after the transform
            // retires it.  Between pieces of one chunk the next piece must have landed: full barrier.
            unsigned long long tp0 = 0;
            if (a.stamps) tp0 = __builtin_amdgcn_s_memrealtime();
            if (pc + 1 == n_pieces && more)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else
                __syncthreads();
            if (a.stamps) acc_pbar += __builtin_amdgcn_s_memrealtime() - tp0;
            cur ^= 1;
        }
        if (more) {
            unsigned long long ta = 0, tb = 0;
            if (a.stamps) ta = __builtin_amdgcn_s_memrealtime();
            if (!(a.dbg & 1)) stage_chunk(ch + 1, true);
            if (a.stamps) tb = __builtin_amdgcn_s_memrealtime();
            __syncthreads();
            if (a.stamps) {
                acc_stage += tb - ta;
                acc_bar += __builtin_amdgcn_s_memrealtime() - tb;
            }
        }
    }

    if (a.stamps) st2 = __builtin_amdgcn_s_memrealtime();
    // ---- epilogue (conv_epilogue.h); accumulators carry the 2^ws weight scale -----------------------
    if (a.dbg & 8) return;
    // (all of the LDS is free after the main loop's last barrier: each wave takes 8.25 KiB of it as the scratch of the
    // statistics reduction; the smallest allocation, BM = 32 without the fixed window pitch, is too small for four of them)
    float2* stat_scr = (BM >= 64 || PF) ? reinterpret_cast<float2*>(smem16) + wave * (32 * 33) : nullptr;
    conv_store_tile<MT, NT, EPI_ROWS, (VT > 1)>(a, acc, a.w_unscale, b, ct * BM + wm * (MT * 32), t0 + wn * (NT * 32), r, h, ncols, Lout,
                            tile_x * WN + wn, stat_scr);
    if (a.stamps && tid == 0) {  // stamps leave through a buffer of their own that nothing else reads
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        unsigned long long* o = a.stamps + (unsigned long long)lin * 8;
        o[0] = st0; o[1] = st1; o[2] = st2;
        __builtin_amdgcn_s_waitcnt(0);
        o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = __builtin_amdgcn_s_getreg(63492);  // HW_REG_HW_ID
        o[5] = __builtin_readcyclecounter() - cyc0;  // shader-clock cycles of this workgroup (o[3] - o[0] is real time)
        o[6] = acc_stage;
        o[7] = acc_bar | (acc_pbar << 32);
    }
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

int conv16_cu_count() {  // of the CURRENT device (cached per device: models on several GPUs launch from several threads)
    if (const int part = cu_count_override()) return part;  // (a CU-partitioned model is launching: its share)
    static std::atomic<int> n[KX_MAX_DEVICES];
    int dev = 0;
    KX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= KX_MAX_DEVICES) return 256;
    int v = n[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t pr;
        KX_HIP(hipGetDeviceProperties(&pr, dev));
        v = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
        n[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int BM, int BN, int WM, int WN, int ACT, int TK, bool PF, int VT = 1>
static void launch_inst16_pf(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    static DynLdsLimit lds_limit;  // raise the dynamic-LDS limit only as far as a launch needs (per device)
    auto kern = conv1d_f16x3_kernel<BM, BN, WM, WN, ACT, TK, PF, VT>;
    const int XW = (BN - 1) * a.stride + (a.K - 1) * a.dil + 1;
    const int XWp = PF ? ((VT > 1) ? 128 : BN + 128) : ((XW + 3) & ~3);  // (as in the kernel)
    KX_REQUIRE(!PF || (a.stride == 1 && XW <= XWp), "conv1d f16x3: the prefetching build needs stride 1 and a window <= BN + 128");
    static const int lds_pad = env_int("KX_LDS_PAD", 0);  // diagnostic: extra LDS to force one workgroup per CU
    size_t lds = 16 * ((size_t)2 * TK * 4 * BM + (size_t)VT * 4 * XWp);
    if (lds + lds_pad <= 160 * 1024) lds += lds_pad;
    KX_REQUIRE(lds <= 160 * 1024, "conv1d f16x3: LDS tile too large for this k/stride");
    KX_REQUIRE(VT == 1 || (a.K == 1 && a.stride == 1 && XW <= 128), "conv1d f16x3: virtual taps need a k=1 GEMM");
    lds_limit.ensure(reinterpret_cast<const void*>(kern), lds);
    dim3 grid((max_cols + BN - 1) / BN, (a.Cout + BM - 1) / BM, a.merge_T > 0 ? 1 : B);
    KX_REQUIRE(grid.x > 0 && grid.y > 0 && grid.y < 65536 && B > 0 && B < 65536, "conv1d f16x3: bad grid");
    KX_REQUIRE(a.merge_T == 0 || (PF && a.K == 1 && a.stride == 1 && a.pad == 0 && !a.in_up2 && !a.nmean &&
                                  a.store != ST_UPSCATTER && !a.stat_part),
               "conv1d f16x3: merged columns need a plain k=1 GEMM");
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, a);
    KX_HIP(hipGetLastError());
}


template <int BM, int BN, int WM, int WN, int ACT, int TK>
static void launch_inst16_act(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    static const int pf_env = env_int("KX_PF", 1);
    const int XW = (BN - 1) * a.stride + (a.K - 1) * a.dil + 1;
    if (pf_env && a.stride == 1 && XW <= BN + 128)
        launch_inst16_pf<BM, BN, WM, WN, ACT, TK, true>(a, B, max_cols, s);
    else
        launch_inst16_pf<BM, BN, WM, WN, ACT, TK, false>(a, B, max_cols, s);
}

template <int BM, int BN, int WM, int WN, int TK>
static void launch_inst16_tk(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    if (a.act == ACT_SNAKE)
        launch_inst16_act<BM, BN, WM, WN, ACT_SNAKE, TK>(a, B, max_cols, s);
    else if (a.act == ACT_LEAKY)
        launch_inst16_act<BM, BN, WM, WN, ACT_LEAKY, TK>(a, B, max_cols, s);
    else
        launch_inst16_act<BM, BN, WM, WN, ACT_NONE, TK>(a, B, max_cols, s);
}

template <int BM, int BN, int WM, int WN>
static void launch_inst16(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    // (weight pieces of 1 and 2 taps were measured too: no faster than 3 anywhere, so only TK = 3 is built)
#ifndef KX_TK
#define KX_TK 3
#endif
    launch_inst16_tk<BM, BN, WM, WN, KX_TK>(a, B, max_cols, s);
}

// Tile of a launch: BN columns per workgroup, WN = waves side by side along the columns.
//   128-row weights: 256 x (2 x 2 waves) by default; 128 x (2 x 2) for short sequences; and for small grids (batch 1:
//   256-column tiles would not even give every CU one workgroup) 128 x (4 x 1): half the tile, but each wave still
//   spans 128 columns, so the fused InstanceNorm partial sums cover the same column groups in the same order as with
//   the 256-wide tile and results stay bit-identical across batch sizes.
// stats: the launch fuses InstanceNorm partial sums into its epilogue.  Their slots must then be 128 columns wide whatever the
// launch geometry (an utterance's sums are added in the same order alone and beside a longer one: batch invariance), so the
// 2 x 2-wave tile of short sequences (64-column slots) is not taken.
void conv16_pick_tile(int BM, int max_cols, int B, int Cout, int K, int dil, int stride, int* bn, int* wn, int ws_force, bool stats,
                      int act, int n_chunks16, int pmode) {
    static const int force = env_int("KX_BN", 0);
    if (ws_force != 1 && act >= 0 && conv16_da_s16_shape(BM, K, dil, stride, act, n_chunks16, false, pmode)) {
        // the S16 form of the direct-A conv: 192 columns on chip-filling grids, 128 on small ones; three / two 64-column slots
        const bool small = (long)((max_cols + 255) / 256) * ((Cout + 127) / 128) * B < 256 && ws_force != 2;
        *bn = small ? 128 : 192;
        *wn = small ? 2 : 3;
        return;
    }
    if (ws_force == 2 && conv16_da_eligible(BM, K, dil, stride, 0)) {  // test hook: the direct-A kernel whatever the grid
        *bn = 256; *wn = 2;
        return;
    }
    if (BM != 128) {
        *bn = 256; *wn = 4;
    } else if ((force == 128 || max_cols <= 160) && !stats) {
        *bn = 128; *wn = 2;
    } else if ((long)((max_cols + 255) / 256) * ((Cout + 127) / 128) * B < 256 && force != 256) {
        *bn = 128; *wn = 1;
    } else {
        *bn = 256; *wn = 2;
    }
}

// The tile width of the launch below when it goes to a kernel that can take a flat tile list (the direct-A conv family), else
// 0.  Must mirror launch_conv1d_f16x3's dispatch exactly (launch_da_inst checks the width the host assumed).
int conv16_flat_bn(const ConvArgs& a, int BM, int B, int max_cols) {
    static const int on = env_int("KX_FLAT", 1);
    if (!on || BM != 128 || B < 2 || max_cols <= 0 || a.merge_T > 0 || a.ws_force == 1 || a.stamps) return 0;
    if (a.K == 1 && a.stride == 1 && !a.stat_part && !a.in_up2 && a.n_chunks16 >= 3 && a.act != ACT_SNAKE) return 0;  // k = 1 GEMM forms
    int bn, wn;
    conv16_pick_tile(BM, max_cols, B, a.Cout, a.K, a.dil, a.stride, &bn, &wn, a.ws_force, a.stat_part != nullptr, a.act, a.n_chunks16,
                     conv16_pmode(a));
    if (conv16_da_s16_shape(BM, a.K, a.dil, a.stride, a.act, a.n_chunks16, a.merge_T > 0, conv16_pmode(a))) return bn;  // 192 / 128
    if (bn == 128 && wn == 1) return conv16_use_da(BM, a.K, a.dil, a.stride, a.merge_T > 0) ? 128 : 0;
    if (bn == 128) return 0;
    return conv16_use_da(BM, a.K, a.dil, a.stride, a.merge_T > 0) ? 256 : 0;
}

void launch_conv1d_f16x3(const ConvArgs& a, int BM, int B, int max_cols, hipStream_t s) {
#ifdef KX_ONLY_MAIN  // (compile-time probe builds: just the dominant instantiation, ~10 s instead of ~3 min)
    launch_inst16_pf<128, 256, 2, 2, ACT_SNAKE, 3, true>(a, B, max_cols, s);
    return;
#else
    KX_REQUIRE(a.n_chunks16 == (a.Cin + CK16 - 1) / CK16 && a.w16 != nullptr, "conv1d f16x3: weights not packed");
    KX_REQUIRE(BM == 128 || a.epi != EPI_GELU_NEW, "conv1d f16x3: gelu epilogue needs the 128-row tile");
    if (max_cols <= 0) return;
    if (BM == 128) {
        // (an 8-wave x 128-register form of the 128x256 tile was tried: it spills and is 6 % slower)
        // (also tried: a 128x192 tile for 3 workgroups per CU: the 168-register cap spills in the main loop, 1.7x slower)
        // k = 1 GEMMs (ALBERT, projections, LSTM input products) take the virtual-tap form; it is also the one
        // kernel whose epilogue carries gelu_new
        if (a.K == 1 && a.stride == 1 && !a.stat_part && !a.in_up2 && a.n_chunks16 >= 3 && a.act != ACT_SNAKE) {
            // (test hook mode 2 keeps the virtual-tap form below, so that the two can be compared bit for bit)
            if (a.ws_force != 1 && conv16_use_dag(a, BM)) {
                launch_conv1d_f16x3_dag(a, B, max_cols, s);
                return;
            }
            // Two 16-channel chunks per super-chunk instead of three: 48 KiB of LDS instead of 73.7, so three workgroups
            // fit a CU (the registers always allowed three) and the 1040 - 1170 workgroups of the ALBERT GEMMs at batch 64
            // run in two rounds of 768 instead of three rounds of 512.  (KX_GEMM_VT=3: the former form.)
            // (small grids -- batch 1 -- keep three: fewer barriers per unit of work, and every workgroup is resident anyway)
            static const int vt_env = env_int("KX_GEMM_VT", 0);
            const long wgs = (long)((max_cols + 127) / 128) * ((a.Cout + 127) / 128);
            const int vt = vt_env ? vt_env : (wgs > 2L * conv16_cu_count() ? 2 : 3);
            if (vt == 3) {
                if (a.act == ACT_LEAKY)
                    launch_inst16_pf<128, 128, 2, 2, ACT_LEAKY, 3, true, 3>(a, B, max_cols, s);
                else
                    launch_inst16_pf<128, 128, 2, 2, ACT_NONE, 3, true, 3>(a, B, max_cols, s);
            } else {
                if (a.act == ACT_LEAKY)
                    launch_inst16_pf<128, 128, 2, 2, ACT_LEAKY, 2, true, 2>(a, B, max_cols, s);
                else
                    launch_inst16_pf<128, 128, 2, 2, ACT_NONE, 2, true, 2>(a, B, max_cols, s);
            }
            return;
        }
        KX_REQUIRE(a.epi != EPI_GELU_NEW, "conv1d f16x3: gelu epilogue exists only for k=1 GEMMs with >= 48 input channels");
        int bn, wn;
        conv16_pick_tile(BM, max_cols, B, a.Cout, a.K, a.dil, a.stride, &bn, &wn, a.ws_force, a.stat_part != nullptr, a.act, a.n_chunks16,
                         conv16_pmode(a));
        if (a.ws_force != 1 && conv16_da_s16_shape(BM, a.K, a.dil, a.stride, a.act, a.n_chunks16, a.merge_T > 0, conv16_pmode(a)))
            launch_conv1d_f16x3_da(a, B, max_cols, s, bn);  // (bn = 192 or 128: the S16 form's tiles)
        else if (bn == 128 && wn == 1 && conv16_use_da(BM, a.K, a.dil, a.stride, a.merge_T > 0) && a.ws_force != 1)
            launch_conv1d_f16x3_da(a, B, max_cols, s, 128);  // (test hook mode 2 keeps the LDS-DMA form for comparison)
        else if (bn == 128 && wn == 1)
            launch_inst16<128, 128, 4, 1>(a, B, max_cols, s);
        else if (bn == 128)
            launch_inst16<128, 128, 2, 2>(a, B, max_cols, s);
        else if (conv16_use_da(BM, a.K, a.dil, a.stride, a.merge_T > 0) && a.ws_force != 1)
            launch_conv1d_f16x3_da(a, B, max_cols, s);
        else
            launch_inst16<128, 256, 2, 2>(a, B, max_cols, s);
    } else if (BM == 64)
        launch_inst16<64, 256, 1, 4>(a, B, max_cols, s);
    else if (BM == 32)
        launch_inst16<32, 256, 1, 4>(a, B, max_cols, s);
    else
        throw Error(1, "conv1d f16x3: unsupported BM");
#endif
}

// ---- weight repacking into split-f16 fragment images -------------------------------------------------
// image: [co_tile][chunk16][tap][hi|lo][h][BM][8] halves
size_t packed_conv16_halves(int rows, int Cin, int K, int BM) {
    const size_t tiles = (rows + BM - 1) / BM, chunks = (Cin + CK16 - 1) / CK16;
    return tiles * chunks * K * 4 * BM * 8;
}

__global__ void absmax_kernel(const float* p, long n, unsigned* out) {
    float m = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(p[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order as uints
}

float device_absmax(const float* p, long n, hipStream_t s) {
    unsigned* d;
    KX_HIP(hipMalloc((void**)&d, 4));
    KX_HIP(hipMemsetAsync(d, 0, 4, s));
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, p, n, d);
    unsigned hbits = 0;
    KX_HIP(hipMemcpyAsync(&hbits, d, 4, hipMemcpyDeviceToHost, s));
    KX_HIP(hipStreamSynchronize(s));
    KX_HIP(hipFree(d));
    float f;
    memcpy(&f, &hbits, 4);
    return f;
}

// power of two that brings max|w| into [2^11, 2^12): lo parts of weights 1000x smaller stay normal
int pick_weight_shift(float absmax) {
    if (!(absmax > 0.f) || !std::isfinite(absmax)) return 0;
    int e;
    frexpf(absmax, &e);  // absmax = m * 2^e, m in [0.5, 1)
    int ws = 12 - e;
    if (ws < -8) ws = -8;
    if (ws > 24) ws = 24;
    return ws;
}

__global__ void pack_conv16_kernel(PackSrc src, _Float16* dst, int Cout, int Cin, int K, int BM, float wscale,
                                   long total_pairs) {
    const int n_chunks = (Cin + CK16 - 1) / CK16;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total_pairs; e += (long)gridDim.x * blockDim.x) {
        long q = e;
        const int j8 = q % 8; q /= 8;
        const int m = q % BM; q /= BM;
        const int hh = q % 2; q /= 2;
        const int tap = q % K; q /= K;
        const int ch = q % n_chunks; q /= n_chunks;
        const int co = (int)q * BM + m, ci = ch * CK16 + hh * 8 + j8;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            int rr = co;
            const float* p = src.p[0];
            if (rr >= src.rows[0]) { rr -= src.rows[0]; p = src.p[1];
                if (rr >= src.rows[1]) { rr -= src.rows[1]; p = src.p[2]; } }
            v = p[((long)rr * Cin + ci) * K + tap] * wscale;
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        // [ct][ch][tap][hl][hh][m][j8]
        const long base = ((((long)q * n_chunks + ch) * K + tap) * 2) * 2 * BM * 8;
        dst[base + ((0 * 2 + hh) * (long)BM + m) * 8 + j8] = hi;
        dst[base + ((1 * 2 + hh) * (long)BM + m) * 8 + j8] = lo;
    }
}

void launch_pack_conv16(const PackSrc& src, void* dst, int Cout, int Cin, int K, int BM, float wscale, hipStream_t s) {
    const long total = (long)packed_conv16_halves(Cout, Cin, K, BM) / 2;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv16_kernel, dim3(blocks), dim3(256), 0, s, src, static_cast<_Float16*>(dst), Cout, Cin, K,
                       BM, wscale, total);
    KX_HIP(hipGetLastError());
}

__global__ void pack_convT16_kernel(const float* w, _Float16* dst, int Cin, int Cout, int sd, int BM, float wscale,
                                    long total_pairs) {
    const int n_chunks = (Cin + CK16 - 1) / CK16;
    const int rows = sd * Cout, k = 2 * sd;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total_pairs; e += (long)gridDim.x * blockDim.x) {
        long q = e;
        const int j8 = q % 8; q /= 8;
        const int m = q % BM; q /= BM;
        const int hh = q % 2; q /= 2;
        const int tap = q % 2; q /= 2;
        const int ch = q % n_chunks; q /= n_chunks;
        const int row = (int)q * BM + m, ci = ch * CK16 + hh * 8 + j8;
        float v = 0.f;
        if (row < rows && ci < Cin) {
            const int sdiv = sd, co = row / sdiv, p = row % sdiv;  // GEMM row = co * s + p (phase fastest: conv_epilogue.h, ST_UPSCATTER)
            v = w[((long)ci * Cout + co) * k + (tap == 0 ? p + sd : p)] * wscale;
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const long base = ((((long)q * n_chunks + ch) * 2 + tap) * 2) * 2 * BM * 8;
        dst[base + ((0 * 2 + hh) * (long)BM + m) * 8 + j8] = hi;
        dst[base + ((1 * 2 + hh) * (long)BM + m) * 8 + j8] = lo;
    }
}

// image layout [..][hi|lo][k-half][BM][8]: blocks of 2 BM 8 halves of hi followed by as many of lo
__global__ void image_to_bf16_kernel(const _Float16* src, unsigned short* dst, long n_halves, int plane) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n_halves; e += (long)gridDim.x * blockDim.x) {
        const long blk = e / (2 * plane), in_blk = e - blk * 2 * plane;
        if (in_blk >= plane) {
            dst[e] = 0;  // (the lo slots: unused by the bf16 form)
            continue;
        }
        const float v = (float)src[e] + (float)src[e + plane];
        unsigned u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);
        dst[e] = (unsigned short)(u >> 16);
    }
}
void launch_image_to_bf16(const void* w16, void* dst, size_t n_halves, hipStream_t s) {
    const int plane = 2 * 128 * 8;  // BM = 128 images only (the direct-A kernels)
    const int blocks = (int)((n_halves + 255) / 256 < 4096 ? (n_halves + 255) / 256 : 4096);
    hipLaunchKernelGGL(image_to_bf16_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const _Float16*>(w16), static_cast<unsigned short*>(dst),
                       (long)n_halves, plane);
    KX_HIP(hipGetLastError());
}

// ---- f16f8 mode: the weights' cross-term operands (layout: kx_common.h, launch_pack_conv8x) -------------------------------
size_t packed_conv8x_bytes(int rows, int Cin, int K) {
    const size_t tiles = (rows + 127) / 128, chunks = (Cin + CK16 - 1) / CK16, groups = (K + 3) / 4;
    return tiles * chunks * groups * 4 * 128 * 32;
}

__global__ void pack_conv8x_kernel(const _Float16* w16, unsigned* dst, int K, long total_dwords) {
    const int NG = (K + 3) / 4;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total_dwords; e += (long)gridDim.x * blockDim.x) {
        long q = e;
        const int d = q % 4; q /= 4;
        const int row = q % 128; q /= 128;
        const int sl = q % 2; q /= 2;
        const int g = q % 4; q /= 4;
        const int mg = q % NG; q /= NG;  // q: row tile x chunk
        const int tap = 4 * mg + 2 * (g >> 1) + sl, hh = g & 1;
        unsigned out = 0u;
        if (tap < K) {
            // split-f16 image [ct][ch][tap][hi|lo][hh][128][8]
            const _Float16* src = w16 + (((q * K + tap) * 2) * 2 * 128) * 8;
            const _Float16* ph = src + ((0 * 2 + hh) * 128L + row) * 8 + 2 * d;
            const _Float16* pl = src + ((1 * 2 + hh) * 128L + row) * 8 + 2 * d;
            const float h0 = __builtin_amdgcn_fmed3f((float)ph[0] * 0.0625f, -448.f, 448.f), h1 = __builtin_amdgcn_fmed3f((float)ph[1] * 0.0625f, -448.f, 448.f);
            const float l0 = __builtin_amdgcn_fmed3f((float)pl[0] * 128.f, -448.f, 448.f), l1 = __builtin_amdgcn_fmed3f((float)pl[1] * 128.f, -448.f, 448.f);
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(l0, l1, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(h0, h1, pk, true);
            out = (unsigned)pk;
        }
        dst[e] = out;
    }
}

void launch_pack_conv8x(const void* w16, void* dst, int rows, int Cin, int K, hipStream_t s) {
    const long total = (long)(packed_conv8x_bytes(rows, Cin, K) / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv8x_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const _Float16*>(w16), static_cast<unsigned*>(dst), K, total);
    KX_HIP(hipGetLastError());
}

void launch_pack_convT16(const float* w, void* dst, int Cin, int Cout, int sd, int BM, float wscale, hipStream_t s) {
    const long total = (long)packed_conv16_halves(sd * Cout, Cin, 2, BM) / 2;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_convT16_kernel, dim3(blocks), dim3(256), 0, s, w, static_cast<_Float16*>(dst), Cin, Cout, sd,
                       BM, wscale, total);
    KX_HIP(hipGetLastError());
}

}  // namespace kx
