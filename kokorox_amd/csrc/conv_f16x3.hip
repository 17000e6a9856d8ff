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
constexpr int TK_MAX = 3;  // taps per weight piece in LDS (template parameter TK <= this)

// sin^2(t) for moderate |t| (snake activations): Cody-Waite reduction to |r| <= pi/4 by multiples of
// pi/2, even Taylor series of sin^2 on the reduced argument, complement on odd quadrants.
// Absolute error ~1e-7 for |t| < 1e3 (tests/test_gpu_kernels.py checks the fused conv against f64).
__device__ __forceinline__ float sin_sq(float t) {
    const float n = rintf(t * 0.636619772367581343f);         // t / (pi/2)
    float r = fmaf(n, -1.57079625129699707031f, t);            // pi/2 high part (exact in 17 bits)
    r = fmaf(n, -7.54978941586159635335e-08f, r);              // pi/2 low part
    const float r2 = r * r;
    float p = fmaf(r2, -4.27561049e-06f, 1.41093474e-04f);     // -2/467775, 2/14175
    p = fmaf(r2, p, -3.17460317e-03f);                         // -1/315
    p = fmaf(r2, p, 4.44444444e-02f);                          // 2/45
    p = fmaf(r2, p, -3.33333333e-01f);                         // -1/3
    p = fmaf(r2, p, 1.0f);
    const float s2 = r2 * p;
    return (((int)n) & 1) ? 1.0f - s2 : s2;
}

// f32 -> (hi, lo) halves, two values packed per dword: hi = f16(v), lo = f16(v - hi)
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi_pk, unsigned& lo_pk) {
    v0 = __builtin_fminf(__builtin_fmaxf(v0, -65504.f), 65504.f);
    v1 = __builtin_fminf(__builtin_fmaxf(v1, -65504.f), 65504.f);
    half2v h2, l2;
    h2[0] = (_Float16)v0;
    h2[1] = (_Float16)v1;
    l2[0] = (_Float16)(v0 - (float)h2[0]);
    l2[1] = (_Float16)(v1 - (float)h2[1]);
    hi_pk = __builtin_bit_cast(unsigned, h2);
    lo_pk = __builtin_bit_cast(unsigned, l2);
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
    const int t0 = blockIdx.x * BN;
    const bool merged = a.merge_T > 0;

    const int Lin = merged ? a.merge_B * a.merge_T : a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = merged ? Lin : a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    const int ncols = (a.store == ST_UPSCATTER) ? (Lin + 1) : Lout;
    if (t0 >= ncols) return;

    const int K = a.K, dil = a.dil, stride = a.stride;
    const int XW = (BN - 1) * stride + (K - 1) * dil + 1;
    const int XWp = (XW + 3) & ~3;
    constexpr int tap_units = 4 * BM;  // uint4 per tap: [hi|lo][h][BM]
    constexpr int piece_units = TK * tap_units;
    // LDS carve (16-B units): two weight-piece buffers, then Xs [hi|lo][octet][XWp]
    uint4* Wbuf = reinterpret_cast<uint4*>(smem16);
    uint4* Xs = Wbuf + 2 * piece_units;
    unsigned* Xs32 = reinterpret_cast<unsigned*>(Xs);

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
        for (int sgm = wave; sgm < nseg; sgm += NWV) glds16(src + sgm * 64 + lane, dst + sgm * 64);
    };

    // ---- input chunk staging.  Wave w takes channel pairs w and w+4 of the chunk's 8 pairs; per-channel
    // AdaIN / snake parameters are wave-uniform, lanes run along time (coalesced), and each lane writes one
    // packed hi pair and one packed lo pair into the [time][8 ch] image.  With PF the raw values of the NEXT
    // chunk are loaded into registers before the MFMA loop of the current one (their latency hides behind
    // the matrix work) and are transformed + written after it.
    constexpr int NI = (VT > 1) ? 2 : 5;  // PF needs XW <= 64 * NI (the launcher checks)
    constexpr int NH = 8 / NWV;            // channel pairs staged per wave
    float raw[VT][NH][NI][2];
    // per-lane element offsets of this lane's NI window columns (they do not depend on the chunk)
    int xoff[NI];
    unsigned okmask = 0;
    if (PF) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int p = p0 + lane + 64 * it;
            if (merged) {
                const int pc = p < Lin ? p : Lin - 1;
                const int bb = pc / a.merge_T, tt = pc - bb * a.merge_T;
                const bool ok = p < Lin && tt < a.in_len.lens[bb];
                okmask |= ok ? (1u << it) : 0u;
                xoff[it] = (int)((long)bb * a.x_bs + tt);
            } else {
                okmask |= (p >= 0 && p < Lin) ? (1u << it) : 0u;
                int pi = up2 ? (p >> 1) : p;
                pi = pi < 0 ? 0 : (pi >= Lsrc ? Lsrc - 1 : pi);
                xoff[it] = pi;
            }
        }
    }
    const float* xbase = merged ? a.x : xb;
    auto load_raw = [&](int sc) {
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            const int ch = sc * VT + vt;  // 16-channel chunk staged as virtual tap vt
#pragma unroll
            for (int half = 0; half < NH; ++half) {
                const int cA = ch * CK16 + 2 * (wave + NWV * half), cB = cA + 1;
                const float* rowA = xbase + (long)(cA < a.Cin ? cA : 0) * a.x_ld;
                const float* rowB = xbase + (long)(cB < a.Cin ? cB : 0) * a.x_ld;
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    raw[vt][half][it][0] = rowA[xoff[it]];
                    raw[vt][half][it][1] = rowB[xoff[it]];
                }
            }
        }
    };
    auto stage_chunk = [&](int sc, bool from_raw) {
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) {
        const int ch = sc * VT + vt;
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            const int pr = wave + NWV * half;  // pair index 0..7 inside the 16-channel chunk
            const int cA = ch * CK16 + 2 * pr, cB = cA + 1;
            const bool okA = cA < a.Cin, okB = cB < a.Cin;
            const int cAc = okA ? cA : 0, cBc = okB ? cB : 0;  // clamped: loads stay in bounds, result masked
            float mA = 0.f, sA = 1.f, hA = 0.f, aA = 1.f, iA = 1.f, mB = 0.f, sB = 1.f, hB = 0.f, aB = 1.f, iB = 1.f;
            if (has_norm) {
                const long nA = (long)b * a.n_bs + cAc, nB = (long)b * a.n_bs + cBc;
                mA = a.nmean[nA]; sA = a.nscale[nA]; hA = a.nshift[nA];
                mB = a.nmean[nB]; sB = a.nscale[nB]; hB = a.nshift[nB];
            }
            if (ACT == ACT_SNAKE) {
                aA = a.alpha[cAc]; iA = 1.0f / aA;
                aB = a.alpha[cBc]; iB = 1.0f / aB;
            }
            const float* rowA = xbase + (long)cAc * a.x_ld;
            const float* rowB = xbase + (long)cBc * a.x_ld;
            // dword index of (hi image, octet g, column 0, channel slot j): ((0*2+g)*XWp)*4 + j/2
            const int g = pr >> 2, jw = pr & 3;
            unsigned* dst_hi = Xs32 + ((vt * 4 + 0 * 2 + g) * XWp) * 4 + jw;
            unsigned* dst_lo = Xs32 + ((vt * 4 + 1 * 2 + g) * XWp) * 4 + jw;
            auto emit = [&](int u, float xA, float xB, bool pok) {
                float yA, yB;
                if (a.dbg & 32) {  // ablation: no activation arithmetic
                    yA = (xA - mA) * sA + hA;
                    yB = (xB - mB) * sB + hB;
                } else {
                    yA = in_act<ACT>((xA - mA) * sA + hA, a.slope, aA, iA);
                    yB = in_act<ACT>((xB - mB) * sB + hB, a.slope, aB, iB);
                }
                yA = (pok && okA) ? yA : 0.f;  // zero padding comes after the activation
                yB = (pok && okB) ? yB : 0.f;
                unsigned hp, lp;
                split_pair(yA, yB, hp, lp);
                if (a.dbg & 64) {  // ablation: no LDS writes (one lane keeps the values alive)
                    if (hp == 0x12345678u && lp == 0x9abcdef0u) dst_hi[0] = hp;
                } else {
                    dst_hi[u * 4] = hp;
                    dst_lo[u * 4] = lp;
                }
            };
            if (PF && from_raw) {
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int u = lane + 64 * it;
                    if (u < XW) emit(u, raw[vt][half][it][0], raw[vt][half][it][1], ((okmask >> it) & 1u) != 0u);
                }
            } else {
                if (PF) {  // first chunk of the prefetching build: same hoisted offsets, loads issued here
#pragma unroll
                    for (int it = 0; it < NI; ++it) {
                        const int u = lane + 64 * it;
                        if (u < XW) emit(u, rowA[xoff[it]], rowB[xoff[it]], ((okmask >> it) & 1u) != 0u);
                    }
                } else {
                    for (int u = lane; u < XW; u += 64) {
                        const int p = p0 + u;
                        int pi = up2 ? (p >> 1) : p;
                        pi = pi < 0 ? 0 : (pi >= Lsrc ? Lsrc - 1 : pi);
                        emit(u, rowA[pi], rowB[pi], p >= 0 && p < Lin);
                    }
                }
            }
        }
      }
    };

    // Co-resident workgroups of one CU start together and would stay in lockstep (staging with staging,
    // MFMA with MFMA, epilogue with epilogue).  Delay the second first-round workgroup of every CU once, so
    // that one block's memory/VALU phases run beside the other's matrix phase; later blocks inherit the offset.
    if (a.stagger_ticks > 0) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const bool delayed = (a.dbg & 16) ? (((lin >> 3) & 1u) != 0u) : (lin >= 256u);
        if (lin < 512u && delayed) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t_start < (unsigned long long)a.stagger_ticks)
                __builtin_amdgcn_s_sleep(32);
        }
    }
    unsigned long long st0 = 0, st1 = 0, st2 = 0, acc_stage = 0, acc_bar = 0;
    if (a.stamps) st0 = __builtin_amdgcn_s_memrealtime();
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
            for (int tt = 0; tt < ((a.dbg & 4) ? 0 : taps); ++tt) {
                const int tap = tp + tt;
                half8 ah[MT], al_[MT], bh[NT], bl[NT];
                const uint4* wt = Wp + tt * tap_units + h * BM + wm * (MT * 32) + r;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    ah[mt] = *reinterpret_cast<const half8*>(&wt[mt * 32]);
                    al_[mt] = *reinterpret_cast<const half8*>(&wt[2 * BM + mt * 32]);
                }
                const uint4* xt = (VT > 1) ? Xs + (tt * 4 + h) * XWp + (wn * (NT * 32) + r) * stride
                                           : Xs + h * XWp + (wn * (NT * 32) + r) * stride + tap * dil;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt] = *reinterpret_cast<const half8*>(&xt[nt * 32 * stride]);
                    bl[nt] = *reinterpret_cast<const half8*>(&xt[2 * XWp + nt * 32 * stride]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
            // Everyone must be done with Wbuf[cur] (and with Xs after the last piece).  After the last piece of a
            // chunk a raw barrier with lgkmcnt(0) only is enough: the transform of the next chunk follows, and the
            // weight piece that was just issued stays in flight across it (a __syncthreads() would drain vmcnt and
            // expose the copy latency right behind a short MFMA burst); the full barrier after the transform
            // retires it.  Between pieces of one chunk the next piece must have landed: full barrier.
            if (pc + 1 == n_pieces && more)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else
                __syncthreads();
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
    conv_store_tile<MT, NT, EPI_ROWS, (VT > 1)>(a, acc, a.w_unscale, b, ct * BM + wm * (MT * 32), t0 + wn * (NT * 32), r, h, ncols, Lout,
                            blockIdx.x * WN + wn);
    if (a.stamps && tid == 0) {  // stamps leave through a buffer of their own that nothing else reads
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        unsigned long long* o = a.stamps + (unsigned long long)lin * 8;
        o[0] = st0; o[1] = st1; o[2] = st2;
        __builtin_amdgcn_s_waitcnt(0);
        o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = __builtin_amdgcn_s_getreg(63492);  // HW_REG_HW_ID
        o[5] = __builtin_amdgcn_s_getreg(63508);  // HW_REG_XCC_ID
        o[6] = acc_stage;
        o[7] = acc_bar;
    }
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <int BM, int BN, int WM, int WN, int ACT, int TK, bool PF, int VT = 1>
static void launch_inst16_pf(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    static size_t lds_limit = 64 * 1024;  // raise the dynamic-LDS limit only as far as a launch needs
    auto kern = conv1d_f16x3_kernel<BM, BN, WM, WN, ACT, TK, PF, VT>;
    const int XW = (BN - 1) * a.stride + (a.K - 1) * a.dil + 1;
    const int XWp = (XW + 3) & ~3;
    const size_t lds = 16 * ((size_t)2 * TK * 4 * BM + (size_t)VT * 4 * XWp);
    KX_REQUIRE(lds <= 160 * 1024, "conv1d f16x3: LDS tile too large for this k/stride");
    KX_REQUIRE(VT == 1 || (a.K == 1 && a.stride == 1 && XW <= 128), "conv1d f16x3: virtual taps need a k=1 GEMM");
    if (lds > lds_limit) {
        KX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_limit = lds;
    }
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
    if (pf_env && XW <= 320)
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
    launch_inst16_tk<BM, BN, WM, WN, 3>(a, B, max_cols, s);
}

int conv16_pick_bn(int BM, int max_cols) {
    static const int force = env_int("KX_BN", 0);
    if (BM == 128 && force == 128) return 128;
    return (BM == 128 && max_cols <= 160) ? 128 : 256;
}

void launch_conv1d_f16x3(const ConvArgs& a, int BM, int B, int max_cols, hipStream_t s) {
    KX_REQUIRE(a.n_chunks16 == (a.Cin + CK16 - 1) / CK16 && a.w16 != nullptr, "conv1d f16x3: weights not packed");
    KX_REQUIRE(BM == 128 || a.epi != EPI_GELU_NEW, "conv1d f16x3: gelu epilogue needs the 128-row tile");
    if (max_cols <= 0) return;
    if (BM == 128) {
        // (an 8-wave x 128-register form of the 128x256 tile was tried: it spills and is 6 % slower)
        // (also tried: a 128x192 tile for 3 workgroups per CU: the 168-register cap spills in the main loop, 1.7x slower)
        // k = 1 GEMMs (ALBERT, projections, LSTM input products) take the virtual-tap form; it is also the one
        // kernel whose epilogue carries gelu_new
        if (a.K == 1 && a.stride == 1 && !a.stat_part && !a.in_up2 && a.n_chunks16 >= 3 && a.act != ACT_SNAKE) {
            if (a.act == ACT_LEAKY)
                launch_inst16_pf<128, 128, 2, 2, ACT_LEAKY, 3, true, 3>(a, B, max_cols, s);
            else
                launch_inst16_pf<128, 128, 2, 2, ACT_NONE, 3, true, 3>(a, B, max_cols, s);
            return;
        }
        KX_REQUIRE(a.epi != EPI_GELU_NEW, "conv1d f16x3: gelu epilogue exists only for k=1 GEMMs with >= 48 input channels");
        if (conv16_pick_bn(BM, max_cols) == 128)
            launch_inst16<128, 128, 2, 2>(a, B, max_cols, s);
        else
            launch_inst16<128, 256, 2, 2>(a, B, max_cols, s);
    } else if (BM == 64)
        launch_inst16<64, 256, 1, 4>(a, B, max_cols, s);
    else if (BM == 32)
        launch_inst16<32, 256, 1, 4>(a, B, max_cols, s);
    else
        throw Error(1, "conv1d f16x3: unsupported BM");
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
            const int p = row / Cout, co = row % Cout;
            v = w[((long)ci * Cout + co) * k + (tap == 0 ? p + sd : p)] * wscale;
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const long base = ((((long)q * n_chunks + ch) * 2 + tap) * 2) * 2 * BM * 8;
        dst[base + ((0 * 2 + hh) * (long)BM + m) * 8 + j8] = hi;
        dst[base + ((1 * 2 + hh) * (long)BM + m) * 8 + j8] = lo;
    }
}

void launch_pack_convT16(const float* w, void* dst, int Cin, int Cout, int sd, int BM, float wscale, hipStream_t s) {
    const long total = (long)packed_conv16_halves(sd * Cout, Cin, 2, BM) / 2;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_convT16_kernel, dim3(blocks), dim3(256), 0, s, w, static_cast<_Float16*>(dst), Cin, Cout, sd,
                       BM, wscale, total);
    KX_HIP(hipGetLastError());
}

}  // namespace kx
