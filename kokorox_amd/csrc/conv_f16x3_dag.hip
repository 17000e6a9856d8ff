// k = 1 GEMMs (ALBERT, projections, LSTM input products; the merged token-axis column space) in the direct-A form of
// conv_f16x3_da.hip: 128 x 128 tile, wave w owns rows [32 w, 32 w + 32) and all 128 columns (1 x 4 accumulator tiles), weight
// fragments from L2 straight into a three-slot register ring, LDS only for the input images.
//
// Why: conv1d_f16x3_kernel's "virtual tap" form of these launches (three 16-channel chunks staged together, weights by
// LDS-DMA) runs at 130 - 210 algorithmic TFLOP/s, 19 % of the matrix pipe: the LDS-DMA copy instructions (see
// conv_f16x3_da.hip), two barriers per 36 MFMAs, and the f16 split of the input done in one piece between the MFMA bursts
// (a k = 1 GEMM has a ninth of a k = 11 conv's matrix work per input element, so that split is as long as the MFMAs).
// Here a super-chunk is three 16-channel chunks = three "taps" reading three separate input images; its 36 MFMAs per
// wave carry, between them, the split of the NEXT super-chunk's 24 elements per lane (two channel pairs behind each of its
// last six column tiles); one
// barrier per super-chunk; three workgroups per CU (48 KiB of LDS, < 168 registers).
// Channels past Cin are staged as zeros and the weight ring re-reads the last real step for them (finite x 0 = 0), so a
// channel count that is not a multiple of 48 needs no special case.
// Per accumulator the products are added in the order of the other kernels (chunks ascending; a_lo b_hi, a_hi b_lo,
// a_hi b_hi): results are bit-identical to the virtual-tap form.
#include "conv_f16x3_common.h"
#include <type_traits>

namespace kx {

template <int I, int N, class F>
__device__ __forceinline__ void static_for_g(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_g<I + 1, N>(f);
    }
}

bool conv16_dag_eligible(const ConvArgs& a, int BM) {
    return BM == 128 && a.K == 1 && a.stride == 1 && !a.stat_part && !a.in_up2 && a.n_chunks16 >= 3 && a.act != ACT_SNAKE &&
           a.nmean == nullptr && a.store != ST_UPSCATTER;
}
bool conv16_use_dag(const ConvArgs& a, int BM) {
    static const int on = getenv("KX_DAG") ? atoi(getenv("KX_DAG")) : 1;
    return on && conv16_dag_eligible(a, BM);
}

template <int ACT>
__global__ __launch_bounds__(256, 3) void conv1d_f16x3_dag_kernel(const ConvArgs a) {
    constexpr int BM = 128, BN = 128, NT = 4, VT = 3;
    constexpr int XWp = BN;                 // columns of one image row
    constexpr int XIMG = 4 * XWp;           // uint4 per image: [hi|lo][octet][XWp]
    constexpr int XBUF = VT * XIMG;         // uint4 per buffer: the three images of a super-chunk
    constexpr int tap_units = 4 * BM;       // uint4 per chunk of the packed weights: [hi|lo][k-half][BM]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    uint4* Xs = reinterpret_cast<uint4*>(smem16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    // XCD-aware tile order, row tile fastest (as conv_f16x3_da.hip)
    int tile_x = blockIdx.x, ct = blockIdx.y;
    {
        const int nx = gridDim.x, ny = gridDim.y, N = nx * ny;
        const int l = blockIdx.x + nx * blockIdx.y;
        int lp = l;
        if (a.xcd_swizzle && N >= 16) {
            const int off = (int)(((long)N * blockIdx.z) & 7);
            const int cls = (l + off) & 7;
            int start = 0;
            for (int c = 0; c < cls; ++c) {
                const int first = (c - off) & 7;
                start += first < N ? (N - first + 7) >> 3 : 0;
            }
            lp = start + (l >> 3);
        }
        if (a.xcd_swizzle) {
            tile_x = lp / ny;
            ct = lp - tile_x * ny;
        }
    }
    const int t0 = tile_x * BN;
    const bool merged = a.merge_T > 0;
    const int Lin = merged ? a.merge_B * a.merge_T : a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = merged ? Lin : a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    const int ncols = Lout;
    if (t0 >= ncols) return;

    const int n_chunks = a.n_chunks16;
    const int n_super = (n_chunks + VT - 1) / VT;
    const float* xb = a.x + (long)b * a.x_bs;

    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;

    // ---- input staging: wave w owns channel octet w & 1 and column block w >> 1 (64 columns) of every chunk: one column
    // per lane, 8 channels per chunk, 24 elements per super-chunk
    const int g = wave & 1, jb = wave >> 1;
    float raw[VT][8];
    int xoff;
    bool pok;
    {
        const int p = t0 - a.pad + lane + 64 * jb;
        if (merged) {
            const int pc = p < Lin ? (p < 0 ? 0 : p) : Lin - 1;
            const int bb = pc / a.merge_T, tt = pc - bb * a.merge_T;
            pok = p >= 0 && p < Lin && tt < a.in_len.lens[bb];
            xoff = (int)((long)bb * a.x_bs + tt);
        } else {
            pok = p >= 0 && p < Lin;
            xoff = p < 0 ? 0 : (p >= Lin ? Lin - 1 : p);
        }
    }
    const float* xbase = merged ? a.x : xb;
    const int cmax_in = a.Cin - 1;
    constexpr int raw_ops = VT * 8;  // vector loads of one load_raw()
    auto load_raw = [&](int sc) __attribute__((always_inline)) {
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = (sc * VT + vt) * CK16 + g * 8 + c;
                raw[vt][c] = xbase[(long)(ci < cmax_in ? ci : cmax_in) * a.x_ld + xoff];
            }
    };
    const float keep = pok ? a.x_prescale : 0.f;
    // one channel pair of one chunk: leaky / identity, zero padding, f16 split, two dwords of the image
    auto xform_pair = [&](const int vt, const int c2, uint4* Xb, int sc) __attribute__((always_inline)) {
        float y2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = 2 * c2 + q;
            const float y = in_act<ACT>(raw[vt][c], a.slope, 1.f, 1.f);
            y2[q] = y * (((sc * VT + vt) * CK16 + g * 8 + c <= cmax_in) ? keep : 0.f);
        }
        unsigned hp, lp;
        split_pair(y2[0], y2[1], hp, lp);
        const int u = lane + 64 * jb;
        unsigned* Xw = reinterpret_cast<unsigned*>(Xb + vt * XIMG);
        Xw[((0 * 2 + g) * XWp + u) * 4 + c2] = hp;
        Xw[((1 * 2 + g) * XWp + u) * 4 + c2] = lp;
    };

    // ---- A ring (inline-asm loads, hand-counted waits: see conv_f16x3_da.hip)
    const uint4* wlane = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_chunks * tap_units + h * BM + wave * 32 + r;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    auto load_A = [&](int s, u32x4& a_hi, u32x4& a_lo) __attribute__((always_inline)) {
        const int sc = s < n_chunks ? s : n_chunks - 1;  // (chunks past Cin meet an all-zero image)
        const uint4* p = wlane + (long)sc * tap_units;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_hi) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_lo) : "v"(p + 2 * BM) : "memory");
    };
    auto wait_A = [&](int age, u32x4& a_hi, u32x4& a_lo) __attribute__((always_inline)) {
        if (age >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if (age >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (age >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (age >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(a_hi), "+v"(a_lo));
    };

    u32x4 ahs[3], als[3];
    load_A(0, ahs[0], als[0]);
    load_A(1, ahs[1], als[1]);
    load_A(2, ahs[2], als[2]);
    load_raw(0);
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) xform_pair(vt, c2, Xs, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (once: the ages below start from an empty queue)
    __syncthreads();
    int ages[3] = {0, 0, 0};
    if (n_super > 1) {
        load_raw(1);
        ages[0] = ages[1] = ages[2] = raw_ops;
    }

    int cur = 0;
    // B fragments: three-entry ring over the 12 column tiles of a super-chunk, read two tiles ahead
    half8 fh[3], fl[3];
    auto load_tile = [&](int vt, int n, half8& fhx, half8& flx) __attribute__((always_inline)) {
        const uint4* xt = Xs + cur * XBUF + vt * XIMG + h * XWp + r + n * 32;
        fhx = *reinterpret_cast<const half8*>(xt);
        flx = *reinterpret_cast<const half8*>(xt + 2 * XWp);
    };
    load_tile(0, 0, fh[0], fl[0]);
    load_tile(0, 1, fh[1], fl[1]);
    constexpr int TILES = NT * VT;  // 12
    for (int sc = 0; sc < n_super; ++sc) {
        const bool more = sc + 1 < n_super;
        static_for_g<0, TILES>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int t = i / NT, n = i % NT, sl = t;  // (VT = 3 = ring depth: tap t always sits in slot t)
            constexpr int e = i % 3, ip = i + 2, e2 = ip % 3;
            if constexpr (n == 0) wait_A(ages[sl], ahs[sl], als[sl]);
            const half8 ah = __builtin_bit_cast(half8, ahs[sl]), al = __builtin_bit_cast(half8, als[sl]);
            if constexpr (ip < TILES) load_tile(ip / NT, ip % NT, fh[e2], fl[e2]);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fh[e], acc[0][n], 0, 0, 0);
            acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fl[e], acc[0][n], 0, 0, 0);
            acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fh[e], acc[0][n], 0, 0, 0);
            // two channel pairs of the NEXT super-chunk's split ride on each of the last six tiles (12 pairs = 3 chunks x 4):
            // the input prefetch was issued at the super-chunk's start and has had six tiles to land.  (In the last
            // super-chunk they run on stale registers into the buffer nobody reads.)
            if constexpr (i >= TILES / 2) {
                constexpr int p0 = 2 * (i - TILES / 2);
                xform_pair(p0 / 4, p0 % 4, Xs + (cur ^ 1) * XBUF, sc + 1);
                xform_pair((p0 + 1) / 4, (p0 + 1) % 4, Xs + (cur ^ 1) * XBUF, sc + 1);
#pragma unroll
                for (int k3 = 0; k3 < 3; ++k3) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);  // its share of the vector work
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (n == NT - 1) {
                // refill slot t with the same tap of the next super-chunk
                if (more) {
                    load_A((sc + 1) * VT + t, ahs[sl], als[sl]);
#pragma unroll
                    for (int o = 0; o < 3; ++o) ages[o] = o == sl ? 0 : ages[o] + 2;
                }
            }
        });
        if (more) {
            // one barrier per super-chunk: the images just written become readable, and every wave has finished reading
            // the other buffer before anybody overwrites it in the NEXT super-chunk
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
            if (sc + 2 < n_super) {
                load_raw(sc + 2);
#pragma unroll
                for (int o = 0; o < 3; ++o) ages[o] += raw_ops;
            }
            load_tile(0, 0, fh[0], fl[0]);
            load_tile(0, 1, fh[1], fl[1]);
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no hand-counted load is in flight past this point)
    conv_store_tile<1, NT, EPI_ROWS, true>(a, acc, a.w_unscale, b, ct * BM + wave * 32, t0, r, h, ncols, Lout, tile_x, nullptr);
}

// ---- the narrow form for small grids: 128 rows x 32 columns per workgroup, no LDS ------------------------------------------
// At batch 1 the ALBERT GEMMs have 128 columns in all: the 128 x 128 tile leaves 6 - 18 workgroups on 256 CUs, each walking
// the whole K axis alone (43 super-chunks with a barrier each for the 768 x 2048 layer: 35 us per launch, 60 launches on the
// forward's critical path).  Here a workgroup takes 32 columns, so four times as many CUs work, and nothing is staged: lane
// (column lane & 31, channel octet lane >> 5) loads the 8 input values of its B fragment itself, applies the same activation
// / padding / f16 split (xform values are those of xform_pair, operation for operation) and feeds the wave's three MFMAs; the
// four waves (32 rows each) repeat that split, which is cheaper than a barrier.  Weights and inputs come through a three-slot
// register ring of inline-asm loads, ten per chunk (2 weight fragments + 8 input values), every one unconditional (chunks past
// the end re-read the last chunk and are multiplied by zero), so the waits are constants: vmcnt(10 (R - 2)), the younger slots.
// Per accumulator the products are added in the order of the other forms: results are bit-identical to them.
// Ring depth R (round 5: 3 -> 6).  A workgroup of this form walks K alone and nothing hides a load but the ring: with three
// slots a chunk's loads had two chunks (~0.35 us of work) to come back from L2 and the walk ran at the latency, 0.69 us per chunk
// (profiles/r05_b1_timeline_before.txt: 33 us for the 48 chunks of the 2304 x 768 launch, 56 such launches on the batch-1
// critical path); six slots give them five.  96 registers of ring: three workgroups per CU instead of four, on grids that have
// at most half a workgroup per CU anyway.
#ifndef KX_DAGN_RING
#define KX_DAGN_RING 6
#endif
template <int ACT>
__global__ __launch_bounds__(256, 3) void conv1d_f16x3_dagn_kernel(const ConvArgs a) {
    constexpr int BM = 128, BN = 32, R = KX_DAGN_RING;
    static_assert(R >= 3 && 10 * (R - 1) <= 63, "ring depth: the waits must fit vmcnt");
    constexpr int tap_units = 4 * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, tile_x = blockIdx.x, ct = blockIdx.y;
    const int t0 = tile_x * BN;
    const bool merged = a.merge_T > 0;
    const int Lin = merged ? a.merge_B * a.merge_T : a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = merged ? Lin : a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    const int ncols = Lout;
    if (t0 >= ncols) return;
    const int n_chunks = a.n_chunks16;
    const int n_super = (n_chunks + R - 1) / R;

    // the lane's input column (a 1-tap GEMM has no padding)
    long xoff;
    bool pok;
    {
        const int p = t0 + r;
        if (merged) {
            const int pc = p < Lin ? p : Lin - 1;
            const int bb = pc / a.merge_T, tt = pc - bb * a.merge_T;
            pok = p < Lin && tt < a.in_len.lens[bb];
            xoff = (long)bb * a.x_bs + tt;
        } else {
            pok = p < Lin;
            xoff = (long)b * a.x_bs + (p >= Lin ? Lin - 1 : p);
        }
    }
    const float keep = pok ? a.x_prescale : 0.f;
    const int cmax_in = a.Cin - 1;
    // byte offsets of the lane's eight values from the first row of a chunk (checked < 2^31 at launch): a chunk's loads take a
    // scalar base and these, no address arithmetic.  The chunks past the end that complete the last round of the ring re-read
    // the last chunk and are multiplied by zero.  A PARTIAL last chunk (Cin not a multiple of 16: the 1090- and 514-channel
    // shortcuts of the decoder; round 5) reads its missing channels from the last real one (an offset set of its own, chosen by
    // the wave-uniform chunk index) and multiplies them by zero.
    unsigned voff[8], voff_last[8];
    unsigned last_ok = 0;  // bit j: channel j of the lane's octet exists in the last chunk
    const int c_last = n_chunks - 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        voff[j] = (unsigned)(((long)(h * 8 + j) * a.x_ld + xoff) * 4);
        const int ci = c_last * CK16 + h * 8 + j;
        const int cl = ci <= cmax_in ? ci : cmax_in;
        last_ok |= ci <= cmax_in ? (1u << j) : 0u;
        voff_last[j] = (unsigned)(((long)(cl - c_last * CK16) * a.x_ld + xoff) * 4);
    }
    const bool partial = (a.Cin % CK16) != 0;
    const uint4* wlane = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_chunks * tap_units + h * BM + wave * 32 + r;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    u32x4 ahs[R], als[R];
    float raw[R][8];
    // chunk c into a slot: ten vector loads, whatever c
    auto load_chunk = [&](int c, u32x4& a_hi, u32x4& a_lo, float (&rw)[8]) __attribute__((always_inline)) {
        const int cc = c < n_chunks ? c : n_chunks - 1;
        const uint4* p = wlane + (long)cc * tap_units;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_hi) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_lo) : "v"(p + 2 * BM) : "memory");
        const float* base = a.x + (long)cc * CK16 * a.x_ld;
        const bool lastp = partial && cc == c_last;  // (wave-uniform)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned vo = lastp ? voff_last[j] : voff[j];
            asm volatile("global_load_dword %0, %1, %2" : "=v"(rw[j]) : "v"(vo), "s"(base) : "memory");
        }
    };
    // (the wait has no operands, a scheduling barrier follows, and only then are the registers handed on: conv_f16x3_da.hip)
    auto wait_chunk = [&](const bool first, u32x4& a_hi, u32x4& a_lo, float (&rw)[8]) __attribute__((always_inline)) {
        // (first: the ring has just been filled, R - 1 younger slots; else the R - 2 slots issued since this one's refill)
        if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 * (R - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 * (R - 2)) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(a_hi), "+v"(a_lo), "+v"(rw[0]), "+v"(rw[1]), "+v"(rw[2]), "+v"(rw[3]), "+v"(rw[4]), "+v"(rw[5]), "+v"(rw[6]),
                     "+v"(rw[7]));
    };
    // the lane's B fragment of chunk c from its eight input values
    auto split8 = [&](int c, const float (&rw)[8], half8& bh, half8& bl) __attribute__((always_inline)) {
        unsigned hp[4], lp[4];
        const float kc = keep * (c < n_chunks ? 1.f : 0.f);
        const unsigned okm = (partial && c >= c_last) ? last_ok : 0xffu;  // (chunks past the end are zeroed by kc already)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            const float y0 = in_act<ACT>(rw[2 * c2], a.slope, 1.f, 1.f) * (((okm >> (2 * c2)) & 1u) ? kc : 0.f);
            const float y1 = in_act<ACT>(rw[2 * c2 + 1], a.slope, 1.f, 1.f) * (((okm >> (2 * c2 + 1)) & 1u) ? kc : 0.f);
            split_pair(y0, y1, hp[c2], lp[c2]);
        }
        bh = __builtin_bit_cast(half8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
        bl = __builtin_bit_cast(half8, make_uint4(lp[0], lp[1], lp[2], lp[3]));
    };

    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
    static_for_g<0, R>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        load_chunk(t, ahs[t], als[t], raw[t]);
    });
    half8 bh, bl;
    wait_chunk(true, ahs[0], als[0], raw[0]);
    split8(0, raw[0], bh, bl);
    for (int sc = 0; sc < n_super; ++sc) {
        static_for_g<0, R>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value, tn = (t + 1) % R;
            const int c = sc * R + t;
            const half8 ah = __builtin_bit_cast(half8, ahs[t]), al = __builtin_bit_cast(half8, als[t]);
            const half8 bh0 = bh, bl0 = bl;
            // The three MFMAs of a chunk are a dependent chain on the one accumulator: each waits for the one before it, and
            // the split of the NEXT chunk's fragment (slot tn: the younger loads are those of the R - 2 slots behind it) fills the gaps.
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, acc[0][0], 0, 0, 0);
            wait_chunk(false, ahs[tn], als[tn], raw[tn]);
            split8(c + 1, raw[tn], bh, bl);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            // slot t is free once its MFMAs are issued (they read ah / al when they issue)
            load_chunk(c + R, ahs[t], als[t], raw[t]);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int sl = 0; sl < R; ++sl)  // (the ring's last loads land in registers that stay reserved until here)
        asm volatile("" ::"v"(ahs[sl]), "v"(als[sl]), "v"(raw[sl][0]), "v"(raw[sl][1]), "v"(raw[sl][2]), "v"(raw[sl][3]), "v"(raw[sl][4]),
                     "v"(raw[sl][5]), "v"(raw[sl][6]), "v"(raw[sl][7]));
    conv_store_tile<1, 1, EPI_ROWS, true>(a, acc, a.w_unscale, b, ct * BM + wave * 32, t0, r, h, ncols, Lout, tile_x, nullptr);
}

template <int ACT>
static void launch_dagn_inst(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    dim3 grid((max_cols + 31) / 32, (a.Cout + 127) / 128, a.merge_T > 0 ? 1 : B);
    KX_REQUIRE(grid.x > 0 && grid.y > 0 && grid.y < 65536 && B > 0 && B < 65536, "conv1d f16x3 dag narrow: bad grid");
    hipLaunchKernelGGL(conv1d_f16x3_dagn_kernel<ACT>, grid, dim3(256), 0, s, a);
    KX_HIP(hipGetLastError());
}

template <int ACT>
static void launch_dag_inst(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    auto kern = conv1d_f16x3_dag_kernel<ACT>;
    constexpr size_t lds = 16 * (size_t)2 * 3 * 4 * 128;  // two buffers of three images: 48 KiB
    dim3 grid((max_cols + 127) / 128, (a.Cout + 127) / 128, a.merge_T > 0 ? 1 : B);
    KX_REQUIRE(grid.x > 0 && grid.y > 0 && grid.y < 65536 && B > 0 && B < 65536, "conv1d f16x3 dag: bad grid");
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    KX_HIP(hipGetLastError());
}

void launch_conv1d_f16x3_dag(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    KX_REQUIRE(conv16_dag_eligible(a, 128), "conv1d f16x3 dag: launch not eligible");
    KX_REQUIRE(a.n_chunks16 == (a.Cin + CK16 - 1) / CK16 && a.w16 != nullptr, "conv1d f16x3 dag: weights not packed");
    KX_REQUIRE(a.pad == 0, "conv1d f16x3 dag: a 1-tap GEMM has no padding");
    if (max_cols <= 0) return;
    // at most half as many 128 x 128 tiles as CUs: the narrow form (KX_DAGN=0: never, 2: always; results are bit-identical either
    // way).  Measured by batch, off / on: 1: 11.85 / 10.55 ms, 4: 16.74 / 15.73, 16: 36.6 / 36.5; at 32 and 64 no launch qualifies
    // (forced everywhere it costs 3 - 5 %: four times the weight traffic and the split repeated by four waves; on the 390-tile grids of the
    // 768-row GEMMs at batch 64 alone it is twice as slow as the 128 x 128 form: 3.09 against 1.59 ms for 12 launches).
    static const int narrow = getenv("KX_DAGN") ? atoi(getenv("KX_DAGN")) : 1;
    const long wgs = (long)((max_cols + 127) / 128) * ((a.Cout + 127) / 128) * (a.merge_T > 0 ? 1 : B);
    // (the narrow form: 32-bit byte offsets into the input)
    const long x_span = ((long)a.x_bs * (a.merge_T > 0 ? a.merge_B : B) + (long)CK16 * a.x_ld) * 4;
    const bool narrow_ok = x_span < (1L << 31);
    if (narrow_ok && (narrow == 2 || (narrow && 2 * wgs <= conv16_cu_count()))) {
        if (a.act == ACT_LEAKY)
            launch_dagn_inst<ACT_LEAKY>(a, B, max_cols, s);
        else
            launch_dagn_inst<ACT_NONE>(a, B, max_cols, s);
        return;
    }
    if (a.act == ACT_LEAKY)
        launch_dag_inst<ACT_LEAKY>(a, B, max_cols, s);
    else
        launch_dag_inst<ACT_NONE>(a, B, max_cols, s);
}

}  // namespace kx
