// conv1d f16x3, "direct A" form of the 128 x 256 tile: the weight fragments go from global memory (L2) straight into
// the registers the MFMAs read them from; LDS holds only the transformed input window (double-buffered).
//
// Why (profiles/r02_weight_stream_ablations.txt): in conv1d_f16x3_kernel the weights reach LDS by LDS-DMA
// (global_load_lds), 24 one-KiB copy instructions per 3-tap piece and workgroup.  Removing those copies (KX_DBG bit 2)
// takes the 128 -> 128, k = 11 launches from 25.7 to 18.0 ms per step; keeping every copy INSTRUCTION but pointing them
// all at one hot KiB (bit 256) leaves 24.8 ms: the cost is the copy instructions, not their bytes.  A CU moves
// ~12 B/clk through LDS-DMA while the fragment reads keep the LDS busy (the producers of a wave-specialised form that was built and dropped in round 2 measured
// ~350 cycles of issue per copy), two workgroups per CU need 24 GB/s of weights, and every copy a wave issues is time it
// cannot issue MFMAs; with the weight stream gone, input staging and epilogue cost 1.1 ms each instead of 6.6 / 6.9.
//
// Here wave w owns output rows [32 w, 32 w + 32) of the 128-row tile and ALL 256 columns (1 x 8 accumulator tiles, the
// same 128 registers as the 2 x 4 split), so the A fragments of a wave are its own: a_hi / a_lo of a (chunk, tap) step
// are one 16-byte global load each per lane, coalesced (the packed image [chunk][tap][hi|lo][k-half][row][8 ch] puts a
// wave's 32 rows x 16 B back to back), issued two to three steps ahead of their use into a three-slot register ring by
// inline asm with hand-counted s_waitcnt vmcnt (see load_A / wait_A for why the compiler cannot be left to count).
// Every weight byte is fetched once per workgroup, as before, but through the vector-load path and with nothing to wait
// for at a barrier: the weight-piece buffers, their four barriers per chunk and the A-operand ds_reads are gone; what is
// left is ONE barrier per 16-channel chunk (the input images alternate between two LDS buffers).
// B fragments: a three-entry ring over the column tiles; a tile's three MFMAs run back to back on its accumulator and the
// fragments of the tile two further on are read under them.
// Five forms of the main loop:
//   * F8 (round 5, the default mode f16f8: every 7- and 11-tap snake conv whose layer carries an 8-bit cross image): the S16 frame
//     below with a_hi b_hi on v_mfma_f32_16x16x32_f16 and the two cross terms of 16 channels x 4 taps on ONE
//     v_mfma_scale_f32_16x16x128_f8f6f4 -- two MFMA-equivalents per product instead of three, ~2^-17 per product instead of 2^-22
//     (see "S16 form, f16f8"); rings and input prefetch are buffer loads, the snake takes sin^2 from v_cos_f32;
//   * S16 (f16x3 mode: snake convs with 11 taps and the un-dilated 7-tap ones; KX_DA_S16=0 switches it off): v_mfma_f32_16x16x32_f16
//     with K = 16 channels x two
//     taps, on a 128 x 192 tile (128 x 128 on small grids) with 64-column statistics slots; the one form whose results are not
//     bit-identical to the others (one instruction sums 32 products) -- its two tile widths are identical to each other;
//   * W2 (the 256-column tile's compile-time tap counts, default): the four waves as 2 x 2, each 64 rows x 128 columns, so that a
//     B fragment read feeds six MFMAs instead of three and the MFMAs of a tile alternate between two accumulators; two ring
//     slots of 16 registers, every vector load unconditional and every wait a compile-time constant (see "W2 form" below;
//     profiles/r03_w2_form.txt).  Bit-identical to the 4 x 1 forms below, which remain for the 128-column tile (small grids),
//     the reduced-precision mode and KX_DA_W2=0;
//   * compile-time tap count (KT = 3, 7, 11: the resblock convs, 85 % of the conv time): a chunk's KT steps are unrolled and
//     the NEXT chunk's input transform (AdaIN affine + snake + f16 split, ~600 vector instructions per wave and chunk) is
//     dealt out in 24 pieces between the MFMAs of the chunk's column tiles, where the matrix pipe hides it; done in one
//     piece at the chunk boundary it costs exactly its own duration (profiles/r02_da_ablations.txt);
//   * run-time tap count (KT = 0: polyphase upsamplers, 1-tap projections): three steps per round, transform at the boundary.
// Per accumulator the products are added in the same order as in conv1d_f16x3_kernel (a_lo b_hi, a_hi b_lo, a_hi b_hi
// per tap, taps and chunks ascending), and the InstanceNorm partial sums cover the same 128-column groups in the same
// order, so results are bit-identical to the other tile shapes (batch invariance, tests/test_gpu_forward.py).
// Eligible launches: 128-row weight tiles, stride 1, window <= 384 columns, not the merged token-axis form.
#include "conv_f16x3_common.h"
#include <type_traits>


#ifndef KX_F8_RAW_AUX
#define KX_F8_RAW_AUX 0  // cache policy of the F8 forms' input prefetch (2 = non-temporal: measured, see profiles/r05_f16f8_form.txt)
#endif
#ifndef KX_F8_HWCOS
#define KX_F8_HWCOS 1  // f16f8 forms: the snake's sin^2 on v_cos_f32 (0: the polynomial of the other forms)
#endif

namespace kx {

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1 (indices that must be constants BEFORE any loop
// is unrolled: register arrays indexed through an unrolled loop variable stayed in scratch in the largest instantiation)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#if !defined(KX_DA_P1) && !defined(KX_DA_W2) && !defined(KX_DA_S16) && !defined(KX_DA_PRE)
bool conv16_da_eligible(int BM, int K, int dil, int stride, int merged) {
    return BM == 128 && stride == 1 && !merged && (K - 1) * dil + 256 <= 384;
}
bool conv16_use_da(int BM, int K, int dil, int stride, int merged) {
    static const int on = getenv("KX_DA") ? atoi(getenv("KX_DA")) : 1;
    return on && conv16_da_eligible(BM, K, dil, stride, merged);
}
#endif

// P1: the opt-in reduced-precision form (KOKOROX_CONV=f16, BASELINE configs[2] "bf16" / the reference's model_fp16
// variants, hf_cache.rs:135-144): ONE v_mfma_f32_32x32x16_f16 per product on the high halves only (weights and activations
// rounded to f16, f32 accumulation); the low halves are neither loaded, nor computed, nor read.
// W2: the unrolled main loop with the four waves as 2 x 2 (64 rows x 128 columns each), see "W2 form" below.
// S16: the unrolled main loop on v_mfma_f32_16x16x32_f16 (K = 32 = the 16 channels of a chunk x TWO taps), see "S16 form" below.
// PRE: the input arrives as a pre-split image (ConvArgs::x16, written once per tensor by split_image_kernel, conv_f16x3_pre.hip:
// AdaIN affine + activation + f16 hi / lo split already applied, in the very layout of the LDS image), so staging a chunk is
// NLD 16-byte loads and NLD 16-byte LDS writes per lane and the transform is gone from this kernel.  For layers whose input
// window is staged by many row tiles (the 1024-row decoder convs: 8, the polyphase upsamplers: 20 / 6), each of which would
// otherwise repeat the transform of the same window (profiles/r05_pmc_sq_*.txt: 9 - 14 vector instructions per MFMA there).
// BF (with P1 only): the reduced-precision form on bf16 instead of f16 (KOKOROX_CONV=bf16: the dtype BASELINE configs[2] names):
// activations rounded to bf16 (round to nearest even) in the staged image, weights from a bf16 image, v_mfma_f32_32x32x16_bf16.
// F8 (with S16 only; KOKOROX_CONV=f16f8, opt-in): the two cross terms a_lo b_hi + a_hi b_lo of a product go to
// v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 images of the four operands (twice the f16 rate: one instruction carries both cross terms
// of 16 channels x 4 taps), a_hi b_hi stays on v_mfma_f32_16x16x32_f16: 2 MFMA-equivalents per product instead of 3, at ~2^-17 per
// product instead of 2^-22 (tools/probes/mfma_f8mix.hip: 1.52 x in a bare loop; see "S16 form, f16f8" below).
//
// Schedule of that loop: which pair-steps carry a cross-term group, and the hand-counted ages of its ring slots.
template <int KT, int RH, int RC, int RAW>
struct F8Sched {
    static constexpr int HS = (KT - 1) / 2, NG = (KT + 1) / 4, NC = 2 * NG;
    // cross-term use c of a super-chunk (c < NG: group c of the even chunk, else group c - NG of the odd chunk) rides on pair-step:
    // the even chunk's on 0, 2, .. (buffer 0, before barrier 2), the odd chunk's on the last NG even ones (buffer 1, behind barrier 1)
    static constexpr int cross_of(int p) {
        if (p & 1) return -1;
        if (p <= 2 * (NG - 1)) return p / 2;
        if (p >= KT - 1 - 2 * (NG - 1)) return NC - 1 - (KT - 1 - p) / 2;
        return -1;
    }
    // vector-memory operations issued between the refill of a slot and the wait that hands it on, in the steady state
    // (kind 0: the hi slot of pair-step id; kind 1: the cross slot of use id).  Program order of a super-chunk: per pair-step
    // [p == HS: the input prefetch] [the waits] .. [hi refill: 2 loads] [cross refill: 4 loads]; the input prefetch behind barrier 3.
    static constexpr int age(int kind, int id) {
        int n = 0, result = -1;
        int st_h[3 * KT + 8] = {}, st_x[3 * NC + 8] = {};
        for (int sc = 0; sc < 2; ++sc) {
            for (int p = 0; p < KT; ++p) {
                if (p == HS) n += RAW;
                if (sc == 1) {
                    if (kind == 0 && p == id) result = n - st_h[KT + p];
                    if (kind == 1 && cross_of(p) == id) result = n - st_x[NC + id];
                }
                n += 2;
                st_h[p + RH <= KT - 1 ? sc * KT + p + RH : (sc + 1) * KT + p % RH] = n;
                const int c = cross_of(p);
                if (c >= 0) {
                    n += 4;
                    st_x[sc * NC + c + RC] = n;
                }
            }
            n += RAW;
        }
        return result < 63 ? result : 63;
    }
    // matrix-pipe time of blocks [first, first + n) of a super-chunk (NB blocks per pair-step) in units of one f16 MFMA pair:
    // 3 with a cross group, 1 without
    static constexpr int cum(int NB, int first, int n) {
        int t = 0;
        for (int r = 0; r < n; ++r) t += cross_of((first + r) / NB) >= 0 ? 3 : 1;
        return t;
    }
    // half-units (of HU) of the transform done before block rel of the phase that starts at block `first` and has NA blocks:
    // proportional to matrix-pipe time, none in the first I0 blocks
    static constexpr int hu_before(int NB, int NA, int HU, int first, int I0, int rel) {
        return rel > I0 ? ((cum(NB, first, rel) - cum(NB, first, I0)) * HU) / (cum(NB, first, NA) - cum(NB, first, I0)) : 0;
    }
};

template <int ACT, int KT, int NTT, bool P1, bool W2 = false, bool S16 = false, bool PRE = false, bool BF = false, bool F8 = false>
__global__ __launch_bounds__(256, (NTT == 8 || S16) ? 2 : 3) void conv1d_f16x3_da_kernel(const ConvArgs a) {
    static_assert(!BF || P1, "BF: a form of the reduced-precision kernels only");
    static_assert(!F8 || (S16 && KT % 4 == 3), "F8: a form of the S16 loop; tap groups of four with one padding slot");
    constexpr bool HWC = F8 && KX_F8_HWCOS;
    static_assert(!PRE || (!P1 && !S16 && ACT == ACT_NONE), "PRE: the activation lives in the image; f16x3 forms on 32x32x16 only");
    static_assert(!W2 || (KT >= 3 && (KT & 1) && NTT == 8), "W2: odd compile-time tap counts on the 256-column tile");
    static_assert(!S16 || (KT >= 3 && (KT & 1) && !P1 && !W2), "S16: odd compile-time tap counts, three MFMAs per product, 4 x 1 waves");
    static_assert(NTT == 8 || NTT == 4 || (NTT == 6 && S16), "tile widths: 256, 128, and 192 columns in the S16 form");
    // NTT = 8: the 128 x 256 tile of chip-filling launches; NTT = 4: 128 x 128 for small grids (batch 1), three workgroups per CU
    constexpr int BM = 128, NT = NTT, BN = 32 * NT;
    // Staged window.  Run-time tap count: BN + 128 columns = NJF 64-column blocks per wave pair, 8 channels each.
    // Compile-time tap counts (the resblock convs, (KT - 1) dil <= 64 checked at launch): BN + 64 columns -- the last block
    // is SPLIT over all four waves by channel (4 of the 16 channels each), so that a lane transforms 8 NJF + 4 elements
    // per chunk instead of 8 (NJF + 1): 20 instead of 24 on the 256-column tile (12 / 16 on the 128-column one), and a
    // sixth less input is fetched (round 2 staged 384 columns for 256 + 10 .. 50 needed).
    constexpr bool W64 = KT > 0;
    // (the 192-column tile of the S16 form: BN + 64 = 256 columns are four WHOLE blocks, two per wave pair, no split block)
    constexpr bool SPLIT = W64 && BN % 128 == 0;
    constexpr int NJF = W64 ? (SPLIT ? BN / 128 : (BN + 64) / 128) : (BN + 128) / 128;  // whole blocks per lane
    constexpr int NJ = NJF + (SPLIT ? 1 : 0);               // + the split block
    constexpr int HU = 8 * NJF + (SPLIT ? 4 : 0);           // elements per lane and chunk = half-units of the interleaved transform
    constexpr int XWp = W64 ? BN + 64 : BN + 128;           // window pitch of one image row (columns)
    static_assert(!W64 || SPLIT || (BN + 64) % 128 == 0, "the window is whole 64-column blocks in pairs, or ends in a split block");
    constexpr int XBUF = 4 * XWp;           // uint4 per input buffer: [hi|lo][octet][XWp]
    constexpr int tap_units = 4 * BM;       // uint4 per (chunk, tap) of the packed weights: [hi|lo][k-half][BM]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    uint4* Xs = reinterpret_cast<uint4*>(smem16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // a pair of transformed values -> the packed operand halves: f16 hi / lo (split_pair), or one bf16 pair (BF: no low part)
    auto pack_pair = [](float v0, float v1, unsigned& hi_pk, unsigned& lo_pk) __attribute__((always_inline)) {
        if constexpr (BF) {
            auto rne = [](float x) __attribute__((always_inline)) {
                unsigned u = __builtin_bit_cast(unsigned, x);
                u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (activations are finite)
                return u >> 16;
            };
            hi_pk = rne(v0) | (rne(v1) << 16);
            lo_pk = 0u;
        } else if constexpr (F8) {
            split_pair_f8(v0, v1, hi_pk, lo_pk);  // (the "lo" planes of the image hold the 8-bit operands of the cross terms)
        } else {
            split_pair(v0, v1, hi_pk, lo_pk);
        }
    };
    // the input activation of the staging paths: in_act, or (f16f8 snake) the hardware-cosine form of xform_a / _b / _c below, operation
    // for operation (a chunk's values must not depend on which path transformed it)
    auto act_in = [](float y, float slope, float al, float ial) __attribute__((always_inline)) {
        if constexpr (ACT == ACT_SNAKE && HWC) {
            const float u = (al * y) * 0.318309886183790672f;
            return __builtin_fmaf(ial, __builtin_fmaf(-0.5f, __builtin_amdgcn_cosf(u), 0.5f), y);
        } else {
            return in_act<ACT>(y, slope, al, ial);
        }
    };
    using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
    // the high-half product of a P1 form: f16 or bf16 operands
    auto mfma_hi = [](const half8& ah, const half8& bh, const f32x16& c) __attribute__((always_inline)) {
        if constexpr (BF)
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    };
    // XCD-aware tile order (as conv1d_f16x3_kernel: workgroups are dealt round-robin over the 8 XCDs by linear block id, and
    // the blocks of one XCD get a contiguous range of tiles so that the window overlap of neighbouring column tiles is an L2
    // hit), extended to the row tiles: the tiles of one utterance are ordered (column tile, row tile) with the ROW tile
    // fastest, so the two row tiles of a 256-channel layer, which stage the same input window, run side by side on one XCD
    // and the second one's input comes from L2 instead of HBM.
    int tile_x = blockIdx.x, ct = blockIdx.y, b = blockIdx.z;
    if (a.tile_prefix) {
        // flat list of a ragged batch: every workgroup is a live tile, and the XCD ranges run over the whole list (in the dense
        // grid every XCD gets the same eighth of every utterance's slab: the ones that hold the tails of shorter utterances
        // run empty and the launch costs as much as B utterances of the longest length)
        const int N = gridDim.x, l = blockIdx.x, ny = a.flat_ny;
        int lp = l;
        if (a.xcd_swizzle && N >= 16) {
            const int cls = l & 7, q = N >> 3, rem = N & 7;  // class c holds q + (c < rem) blocks
            lp = cls * q + (cls < rem ? cls : rem) + (l >> 3);
        }
        const int gt = lp / ny;  // global column tile
        ct = lp - gt * ny;
        int lo = 0, hi = a.flat_B;  // the utterance with tile_prefix[b] <= gt < tile_prefix[b + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.tile_prefix[mid] <= gt) lo = mid;
            else hi = mid;
        }
        b = __builtin_amdgcn_readfirstlane(lo);
        tile_x = gt - a.tile_prefix[b];
    } else {
        const int nx = gridDim.x, ny = gridDim.y, N = nx * ny;
        const int l = blockIdx.x + nx * blockIdx.y;  // linear id inside the utterance's slab (dispatch order)
        int lp = l;
        if (a.xcd_swizzle && N >= 16) {
            const int off = (int)(((long)N * blockIdx.z) & 7);  // XCD class of l = 0 in this slab
            const int cls = (l + off) & 7;
            int start = 0;
            for (int c = 0; c < cls; ++c) {
                const int first = (c - off) & 7;  // smallest l of class c
                start += first < N ? (N - first + 7) >> 3 : 0;
            }
            lp = start + (l >> 3);  // l = first_cls + 8 j is the j-th block of its class: j = l >> 3 (first < 8)
        }
        if (a.xcd_swizzle) {
            tile_x = lp / ny;
            ct = lp - tile_x * ny;
        }
    }
    // De-phasing (a.dephase_cycles > 0): workgroups of one launch all take the same time per tile, so the workgroups that
    // start together reach their epilogues together and the chip alternates between a phase with the matrix pipes busy
    // and HBM idle and one with every CU storing at once.  A chosen half of the FIRST round of workgroups starts half a
    // tile late (once per launch); the offset then persists from tile to tile.
    if (a.dephase_cycles > 0) {
        const unsigned gl = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);  // dispatch order
        const bool late = a.dephase_mode == 1 ? (gl < 512u && ((gl >> 3) & 1u)) : (gl >= 256u && gl < 512u);
        if (late) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime() + (unsigned long long)a.dephase_cycles;
            while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(64);
        }
    }
    const int t0 = tile_x * BN;
    const int Lin = a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    const int ncols = (a.store == ST_UPSCATTER) ? (Lin + 1) : Lout;
    if (t0 >= ncols) return;

    const int K = KT > 0 ? KT : a.K, dil = a.dil;
    const int n_chunks = a.n_chunks16;
    const int n_steps = n_chunks * K;
    const float* xb = a.x + (long)b * a.x_bs;
    const int p0 = t0 - a.pad;
    const bool has_norm = a.nmean != nullptr;
    const int up2 = a.in_up2;
    const int Lsrc = up2 ? ((Lin + 1) >> 1) : Lin;

    f32x16 acc[1][NT];  // (W2: tiles 0 .. NT/2 - 1 = row block 0, the rest = row block 1 of the wave's 64 x 128)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
    constexpr int NB16 = S16 ? 2 * NT : 1;  // S16: 2 row blocks x 2 NT column blocks of 16 x 16 (the same registers)
    f32x4v acc16[2][NB16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB16; ++j) acc16[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // ---- input chunk staging: the scheme of conv1d_f16x3_kernel (wave w: channel octet w & 1, every other 64-column
    // block of the window; raw values and per-channel parameters of the NEXT chunk prefetched across the MFMA loop)
    const int g = wave & 1, jb = wave >> 1;
    float raw[NJ][8];  // (split block: elements 0..3 = channels 4 jb .. 4 jb + 3 of the wave's octet)
    float praw[4];
    int xoff[NJ];
    unsigned okmask = 0;
    // first column of the wave's block j: whole blocks alternate between the wave pairs, the split block is the window's last
    auto blk_col = [&](int j) __attribute__((always_inline)) { return (SPLIT && j == NJF) ? 64 * 2 * NJF : 64 * (jb + 2 * j); };
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = p0 + lane + blk_col(j);
        okmask |= (p >= 0 && p < Lin) ? (1u << j) : 0u;
        int pi = up2 ? (p >> 1) : p;
        pi = pi < 0 ? 0 : (pi >= Lsrc ? Lsrc - 1 : pi);
        xoff[j] = pi;
    }
    // PRE: lane t moves the uint4 t + 256 i (i < NLD) of a chunk's [hi|lo][octet][XWp] image from global memory to LDS: plane
    // = index / XWp, window column = index % XWp; the byte offset inside a chunk of the global image and the in-range flag do not
    // depend on the chunk (columns outside [0, Lin) are zero padding: loaded from a clamped address, written as zeros)
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    constexpr int NLD = PRE ? (4 * XWp) / 256 : 1;
    static_assert(!PRE || (4 * XWp) % 256 == 0, "PRE: whole rounds of 256 lanes");
    u32x4 rawq[NLD];
    unsigned qoff[NLD];
    unsigned okq = 0;
    const buf_rsrc xq = make_buf(PRE ? reinterpret_cast<const char*>(a.x16) + (long)b * a.x16_bs : reinterpret_cast<const char*>(a.x));
    const unsigned chunk_bytes = PRE ? 64u * (unsigned)a.x16_ld : 0u;  // four planes of x16_ld columns x 16 B
    const buf_rsrc xrow = make_buf(xb);  // (f16f8 forms: the utterance's rows; an utterance's tensor is < 4 GiB, checked at launch)
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i, plane = idx / XWp, u = idx - plane * XWp;
            const int p = p0 + u;
            okq |= (p >= 0 && p < Lin) ? (1u << i) : 0u;
            const int pc = p < 0 ? 0 : (p >= Lin ? Lin - 1 : p);
            qoff[i] = 16u * (unsigned)(plane * a.x16_ld + pc);
        }
    }
    const int cmax_in = a.Cin - 1;
    // (the three InstanceNorm parameter loads are UNCONDITIONAL -- without a norm they read the first input row and are
    // discarded -- so that a prefetch batch is the same number of vector loads whatever the launch: wait_A counts them, and
    // tests/test_asm_audit_cpu.py checks the count in the generated code)
    const float* nm_src = has_norm ? a.nmean + (long)b * a.n_bs : xb;
    const float* ns_src = has_norm ? a.nscale + (long)b * a.n_bs : xb;
    const float* nh_src = has_norm ? a.nshift + (long)b * a.n_bs : xb;
    auto load_params = [&](int ch, float (&pv)[4]) __attribute__((always_inline)) {
        const int c = ch * CK16 + g * 8 + (lane & 7);
        const int cc = c < cmax_in ? c : cmax_in;
        const float m = nm_src[cc], sc = ns_src[cc], sh = nh_src[cc];
        // (channels past the end of a partial last chunk get scale = shift = 0: their transformed value is exactly 0 through
        // every activation, so the per-element code needs no channel test)
        const bool cvalid = c <= cmax_in;
        pv[0] = has_norm ? m : 0.f;
        pv[1] = cvalid ? (has_norm ? sc : 1.f) : 0.f;
        pv[2] = (cvalid && has_norm) ? sh : 0.f;
        pv[3] = (ACT == ACT_SNAKE) ? a.alpha[cc] : 1.f;  // (its reciprocal is taken when the chunk is transformed: a
                                                          // division here would wait for the load on the spot)
    };
    auto load_raw = [&](int ch) __attribute__((always_inline)) {
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                rawq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xq, qoff[i], (unsigned)ch * chunk_bytes, 0));
            return;
        }
        load_params(ch, praw);
        if constexpr (F8) {
            // f16f8 forms: buffer loads -- the utterance's descriptor in scalar registers, the lane's column as a 32-bit offset, the
            // channel's row as a scalar offset: no 64-bit vector address per load and no address registers (the 192-column tile has
            // none to spare)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = ch * CK16 + g * 8 + c;
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(ci < cmax_in ? ci : cmax_in) * (4u * (unsigned)a.x_ld);
#pragma unroll
                for (int j = 0; j < NJF; ++j) raw[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrow, 4u * (unsigned)xoff[j], so, KX_F8_RAW_AUX));
            }
            if constexpr (SPLIT) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ci = ch * CK16 + g * 8 + jb * 4 + c;
                    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(ci < cmax_in ? ci : cmax_in) * (4u * (unsigned)a.x_ld);
                    raw[NJF][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrow, 4u * (unsigned)xoff[NJF], so, KX_F8_RAW_AUX));
                }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ci = ch * CK16 + g * 8 + c;
            const float* row = xb + (long)(ci < cmax_in ? ci : cmax_in) * a.x_ld;
#pragma unroll
            for (int j = 0; j < NJF; ++j) raw[j][c] = row[xoff[j]];
        }
        if constexpr (SPLIT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ci = ch * CK16 + g * 8 + jb * 4 + c;
                raw[NJF][c] = (xb + (long)(ci < cmax_in ? ci : cmax_in) * a.x_ld)[xoff[NJF]];
            }
        }
    };
    struct Oct { float m[8], s[8], h[8], al[8], ial[8]; };
    auto unpack_params = [&](const float (&pv)[4], Oct& o) __attribute__((always_inline)) {
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
    auto emit8 = [&](uint4* Xb, int ch, int u, const float (&x8)[8], const Oct& o, bool pok) __attribute__((always_inline)) {
        unsigned hp[4], lp[4];
        const float keep = pok ? a.x_prescale : 0.f;  // (zero padding and the activation pre-scale in one multiply)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            float y2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * c2 + q;
                const float y = act_in(__builtin_fmaf(x8[c] - o.m[c], o.s[c], o.h[c]), a.slope, o.al[c], o.ial[c]);  // (explicit fma: see conv_epilogue.h)
                y2[q] = y * keep;
            }
            pack_pair(y2[0], y2[1], hp[c2], lp[c2]);
        }
        Xb[(0 * 2 + g) * XWp + u] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        if constexpr (!P1) Xb[(1 * 2 + g) * XWp + u] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
    };
    // the split block's four elements (channels 4 jb + c of the octet; parameters by run-time lane index: jb is wave-uniform)
    auto emit4 = [&](uint4* Xb, int ch, int u, const float (&x8)[8], const float (&pv)[4], bool pok) __attribute__((always_inline)) {
        const float al_or_rcp = (lane & 8) ? 1.0f / pv[3] : pv[3];
        const float keep = pok ? a.x_prescale : 0.f;
        unsigned hp[2], lp[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            float y2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * c2 + q, cl = jb * 4 + c;
                const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[0]), cl));
                const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[1]), cl));
                const float sh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[2]), cl));
                const float al = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_or_rcp), cl));
                const float ial = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_or_rcp), cl + 8));
                const float y = act_in(__builtin_fmaf(x8[c] - m, sc, sh), a.slope, al, ial);
                y2[q] = y * keep;
            }
            pack_pair(y2[0], y2[1], hp[c2], lp[c2]);
        }
        uint2* Xw2 = reinterpret_cast<uint2*>(Xb);
        Xw2[((0 * 2 + g) * XWp + u) * 2 + jb] = make_uint2(hp[0], hp[1]);
        if constexpr (!P1) Xw2[((1 * 2 + g) * XWp + u) * 2 + jb] = make_uint2(lp[0], lp[1]);
    };
    // PRE: unit i of a chunk's staging = one 16-byte LDS write (conflict-free: consecutive lanes, consecutive 16 B)
    auto pre_unit = [&](const int i, uint4* Xb) __attribute__((always_inline)) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        const u32x4 v = ((okq >> i) & 1u) ? rawq[i] : z;
        Xb[tid + 256 * i] = __builtin_bit_cast(uint4, v);
    };
    auto stage_from_raw = [&](uint4* Xb, int ch) __attribute__((always_inline)) {
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) pre_unit(i, Xb);
            return;
        }
        Oct o;
        unpack_params(praw, o);
#pragma unroll
        for (int j = 0; j < NJF; ++j) emit8(Xb, ch, lane + blk_col(j), raw[j], o, ((okmask >> j) & 1u) != 0u);
        if constexpr (SPLIT) emit4(Xb, ch, lane + blk_col(NJF), raw[NJ - 1], praw, ((okmask >> NJF) & 1u) != 0u);
    };

    // ---- the same transform cut into HU / 2 units of one column block x one channel pair (2 of the lane's HU elements),
    // each in two halves of one element (~25 vector instructions; the second half packs the pair and writes its two
    // dwords of the image), to be issued BETWEEN the MFMAs of 24 consecutive column tiles = three steps.
    // Measured (profiles/r02_da_ablations.txt): the transform of a chunk, done in one piece at the chunk's end, costs
    // exactly its own duration (7.2 of 24.8 ms on the k = 11 launches; dropping it, loads kept: 17.7 ms): the partner
    // workgroup's MFMAs do not fill the gap (two co-resident workgroups that start together stay in phase).  A wave's own
    // MFMAs do: the matrix pipe runs an MFMA for 32 cycles while the wave's vector instructions keep issuing.
#ifdef KX_DA_NO_XFORM
    auto keep_elem = [&](const int hh) __attribute__((always_inline)) {
        const int U = hh / 2, half = hh & 1;
        const bool split = SPLIT && U >= 4 * NJF;
        const int j = split ? NJF : U % NJF, cr = split ? 2 * (U - 4 * NJF) + half : 2 * (U / NJF) + half;
        asm volatile("" ::"v"(raw[j][cr]), "v"(praw[0]), "v"(praw[1]), "v"(praw[2]), "v"(praw[3]));
    };
#endif
    float al_rcp_x = 1.f;  // (alpha | 1 / alpha by lane, set by the first part of a chunk's transform)
    float y_carry = 0.f;   // (first element of a pair, from half 0 to half 1)
    float xt_ = 0.f, xz_ = 0.f;  // (state of the element in flight: the affine value, then z = r^2 / sin^2)
    // One element of the transform in three parts, placed behind the three MFMAs of a column tile.  The arithmetic is that
    // of emit8 / in_act / sin_sq, operation for operation (results must not depend on where a chunk was transformed).
    // U: unit (whole blocks: channel pair U / NJF of column block U % NJF, so that consecutive units are the SAME channel pair in
    // the wave's blocks and share their per-channel parameters -- five v_readlane per element otherwise), half: element of the
    // pair; both constants once inlined.
    auto xform_a = [&](const int U, const int half, int ch) __attribute__((always_inline)) {
        // units 0 .. 4 NJF - 1: whole blocks (block U % NJF, channel pair U / NJF); the last two: the split block's two pairs
        const bool split = SPLIT && U >= 4 * NJF;
        const int j = split ? NJF : U % NJF, cr = split ? 2 * (U - 4 * NJF) + half : 2 * (U / NJF) + half;  // element in raw[j]
        const int cq = split ? jb * 4 + cr : cr;                                                         // channel of the octet
        if (U == 0 && half == 0) al_rcp_x = (lane & 8) ? 1.0f / praw[3] : praw[3];
        const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, praw[0]), cq));
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, praw[1]), cq));
        const float sh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, praw[2]), cq));
        xt_ = __builtin_fmaf(raw[j][cr] - m, sc, sh);
        // (an opaque hand-over: two half-units on one tile otherwise get SLP-packed into v_pk_add_f32 / v_pk_fma_f32, which
        // are slow beside MFMAs on gfx950 -- seen in the generated code of the leaky k = 3 form)
        asm volatile("" : "+v"(xt_));
        if (ACT == ACT_SNAKE && HWC) {
            // f16f8: sin^2 t = 0.5 - 0.5 cos 2t on the hardware cosine (v_cos_f32 takes turns: 2t / 2 pi), max error 3.1e-6 absolute
            // (tools/probes/sin_accuracy.hip) -- under this mode's 1e-5 per product -- for 5 vector-issue slots instead of 12
            const float al = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_rcp_x), cq));
            xz_ = (al * xt_) * 0.318309886183790672f;
        } else if (ACT == ACT_SNAKE) {
            const float al = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_rcp_x), cq));
            const float t = al * xt_;
            const float n = rintf(t * 0.318309886183790672f);
            float rr = fmaf(n, -3.14159274101257324f, t);
            rr = fmaf(n, 8.74227765734758578e-08f, rr);
            xz_ = rr * rr;
        }
    };
    auto xform_b = [&]() __attribute__((always_inline)) {
        if (ACT == ACT_SNAKE && HWC) {
            xz_ = __builtin_fmaf(-0.5f, __builtin_amdgcn_cosf(xz_), 0.5f);
        } else if (ACT == ACT_SNAKE) {
            const float z = xz_;
            float pp = fmaf(z, -3.6197402550897095e-06f, 1.3928599946666651e-04f);
            pp = fmaf(z, pp, -3.1722760759294033e-03f);
            pp = fmaf(z, pp, 4.4443082064390182e-02f);
            pp = fmaf(z, pp, -3.3333304524421692e-01f);
            pp = fmaf(z, pp, 1.0f);
            xz_ = z * pp;
        }
    };
    auto xform_c = [&](const int U, const int half, uint4* Xb, int ch) __attribute__((always_inline)) {
        const bool split = SPLIT && U >= 4 * NJF;
        const int j = split ? NJF : U % NJF;
        const int c2 = split ? jb * 2 + (U - 4 * NJF) : U / NJF;  // channel pair of the octet = dword of the 16-byte slot
        const int cq = 2 * c2 + half;
        float y;
        if (ACT == ACT_SNAKE) {
            const float ial = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_rcp_x), cq + 8));
            y = fmaf(ial, xz_, xt_);
        } else {
            y = in_act<ACT>(xt_, a.slope, 1.f, 1.f);
        }
        const float keep = ((okmask >> j) & 1u) ? a.x_prescale : 0.f;
        y = y * keep;
        if (half == 0) {
            y_carry = y;
        } else {
            unsigned hp, lp;
            pack_pair(y_carry, y, hp, lp);
            const int u = lane + blk_col(j);
            unsigned* Xw = reinterpret_cast<unsigned*>(Xb);
            Xw[((0 * 2 + g) * XWp + u) * 4 + c2] = hp;
            if constexpr (!P1) Xw[((1 * 2 + g) * XWp + u) * 4 + c2] = lp;
        }
    };

    // ---- A fragments: lane (row r, k-half h) of wave w reads 16 B at [step][hi|lo][h][32 w + r].
    // The loads are inline asm with hand-counted waits.  Left to the compiler, the ring (a loop-carried set of pending
    // loads whose age differs between the paths into the loop header) gets an s_waitcnt vmcnt(0) at every loop header:
    // the youngest weight load AND the chunk's input prefetch drained once per three steps.  Vector-memory operations of
    // a wave complete in order, so "all but the N youngest are done" is exact once N is known: age[i] counts the
    // operations issued after slot i's refill (2 per refill of another slot, raw_ops per input prefetch batch); the
    // compiler's own waits for its own loads stay safe, they can only over-wait when these loads sit among theirs.
    const uint4* wlane = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_steps * tap_units + h * BM + wave * 32 + r;
    auto load_A = [&](int s, u32x4& a_hi, u32x4& a_lo) __attribute__((always_inline)) {
        const int sc = s < n_steps ? s : n_steps - 1;  // (only the three loads of the prologue can point past the end)
        const uint4* p = wlane + (long)sc * tap_units;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_hi) : "v"(p) : "memory");
        if constexpr (!P1) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_lo) : "v"(p + 2 * BM) : "memory");
    };
    constexpr int APL = P1 ? 1 : 2;  // vector loads per ring refill (what the hand-counted waits add per refill)
    // wait until at most `age` (rounded down to a value this switch knows) vector-memory operations are outstanding.
    // The compiler takes the result of a load asm for complete the moment the asm is issued, so nothing may touch the
    // fragments before this wait: the wait has no operands (a tied "+v" operand made the register allocator copy the
    // still-empty registers elsewhere BEFORE the wait: stale fragments whenever the memory system was slow), a scheduling
    // barrier follows it, and only then does an empty asm hand the fragments on as new values: any copy the allocator
    // wants is made from that point, after the data has landed.
    auto wait_vm = [&](int age) __attribute__((always_inline)) {
        if (age >= 60) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
        else if (age >= 34) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
        else if (age >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (age >= 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        else if (age >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if (age >= 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
        else if (age >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (age >= 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        else if (age >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (age >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (age >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (age >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (age >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto wait_A = [&](int age, u32x4& a_hi, u32x4& a_lo) __attribute__((always_inline)) {
        wait_vm(age);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P1) asm volatile("" : "+v"(a_hi));
        else asm volatile("" : "+v"(a_hi), "+v"(a_lo));
    };
    // S16: a ring slot holds the fragments of a PAIR of steps for the wave's two 16-row blocks: lane (row lane & 15, k-group
    // lane >> 4) takes k-groups 0, 1 = the two channel octets of the pair's first step, 2, 3 = those of its second step, so a
    // lane reads the same packed image at [step + (lane >> 5)][hi|lo][(lane >> 4) & 1][32 w + 16 i + (lane & 15)]: four loads.
    const int r16 = lane & 15, tau16 = lane >> 5, h16 = (lane >> 4) & 1;
    const uint4* wlane16 = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_steps * tap_units + (long)tau16 * tap_units + h16 * BM + wave * 32 + r16;
    auto load_A16 = [&](int s0, u32x4 (&xh)[2], u32x4 (&xl)[2]) __attribute__((always_inline)) {
        // (unconditional, so that the waits are compile-time constants: a pair past the end re-reads the last one, and the loop's
        // exit keeps the registers reserved until the final wait)
        const uint4* p = wlane16 + (long)(s0 < n_steps - 2 ? s0 : n_steps - 2) * tap_units;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xh[0]) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(xh[1]) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xl[0]) : "v"(p + 2 * BM) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(xl[1]) : "v"(p + 2 * BM) : "memory");
    };
    // W2: wave (wr, wc) = (wave >> 1, wave & 1) owns rows [64 wr, 64 wr + 64) and columns [128 wc, 128 wc + 128): a ring slot
    // holds the fragments of a step for its TWO 32-row blocks (the same packed image, rows 64 wr + 32 i + r).
    const int wr2 = wave >> 1, wc2 = wave & 1;
    const uint4* wlane2 = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_steps * tap_units + h * BM + wr2 * 64 + r;
    constexpr int APL2 = P1 ? 2 : 4;  // vector loads per refill of a W2 slot
    auto load_A2 = [&](int s, u32x4 (&xh)[2], u32x4 (&xl)[2]) __attribute__((always_inline)) {
        // (unconditional, so that the waits of the W2 loop are compile-time constants: a step past the end re-reads the last one,
        // and the loop's exit keeps the registers reserved until the final wait)
        const uint4* p = wlane2 + (long)(s < n_steps ? s : n_steps - 1) * tap_units;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xh[0]) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(xh[1]) : "v"(p) : "memory");
        if constexpr (!P1) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xl[0]) : "v"(p + 2 * BM) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(xl[1]) : "v"(p + 2 * BM) : "memory");
        }
    };
    auto wait_A2 = [&](int age, u32x4 (&xh)[2], u32x4 (&xl)[2]) __attribute__((always_inline)) {
        wait_vm(age);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P1) asm volatile("" : "+v"(xh[0]), "+v"(xh[1]));
        else asm volatile("" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xl[0]), "+v"(xl[1]));
    };
    // vector loads of one load_raw(): exact counting is worth 2 % of the step against counting the asm loads only (always
    // a safe under-estimate: 123.8 vs 126.2 ms)
    constexpr int raw_ops = PRE ? NLD : HU + 3 + (ACT == ACT_SNAKE ? 1 : 0);
    constexpr int HUX = PRE ? NLD : HU;   // units of a chunk's staging dealt out between the MFMAs of the unrolled forms
    constexpr int UVI = PRE ? 5 : 27;     // vector instructions of one unit (what the scheduling pipelines are sized for)
    // Three-deep A ring with STATIC slots: step s uses slot s % 3 and refills it for step s + 3 as soon as its MFMAs are
    // issued, so a fragment is requested two whole steps before its use and nothing ever moves between registers.  The
    // (chunk, tap) walk is flattened and unrolled by three for that.
#ifdef KX_DA_STAMPS  // diagnostic build only (tools/stamp_timeline.py): per-workgroup phase stamps; nothing reads them back
    unsigned long long st0 = 0, st1 = 0, st2 = 0, cyc0 = 0, acc_bar = 0, acc_wait = 0;
    if (a.stamps) {
        st0 = __builtin_amdgcn_s_memrealtime();
        cyc0 = __builtin_readcyclecounter();
    }
#endif
    u32x4 ah0 = {0, 0, 0, 0}, al0 = {0, 0, 0, 0}, ah1 = {0, 0, 0, 0}, al1 = {0, 0, 0, 0}, ah2 = {0, 0, 0, 0}, al2 = {0, 0, 0, 0};
    u32x4 a2h[2][2] = {}, a2l[2][2] = {};  // W2 / S16: [slot][row block]
    if constexpr (W2) load_A2(0, a2h[1], a2l[1]);  // (a chunk's first step arrives in slot 1, see below)
    if constexpr (S16 && !F8) load_A16(0, a2h[1], a2l[1]);  // (a super-chunk's first pair likewise)
    if constexpr (!W2 && !S16) {
    load_A(0, ah0, al0);
    load_A(1, ah1, al1);
    // (a slot that is never consumed must never be loaded: its asm result is dead to the compiler, which hands the registers
    // to something else -- addresses -- while the load is still on its way.  That was a memory fault in a compile-time
    // two-tap form of the polyphase upsamplers, which was correct once guarded like this but no faster than the run-time
    // form -- 2.69 / 1.96 ms against 2.66 / 1.90 -- and is not built.)
    if constexpr (KT == 0 || KT >= 3) load_A(2, ah2, al2);
    else ah2 = al2 = u32x4{0, 0, 0, 0};
    }

    load_raw(0);
    stage_from_raw(Xs, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (once: the ages below start from an empty queue)
    __syncthreads();
#ifdef KX_DA_STAMPS
    if (a.stamps) st1 = __builtin_amdgcn_s_memrealtime();
#endif
    int age0 = 0, age1 = 0, age2 = 0;
    if (n_chunks > 1 && !(a.dbg & 1)) {
        load_raw(1);
        age0 = age1 = age2 = raw_ops;
    }

    if constexpr (S16 && F8) {
        // ================= S16 form, f16f8: a_hi b_hi on v_mfma_f32_16x16x32_f16, the cross terms on the 8-bit scaled MFMA =========
        // The frame is the S16 form's below (super-chunks of two chunks, KT pair-steps, the three barriers, the transform dealt out
        // behind the MFMAs); what changes:
        //   * a pair-step's blocks run TWO f16 MFMAs (a_hi b_hi of the wave's two row blocks) instead of six;
        //   * the cross terms of a chunk (16 channels x KT taps) are NG = (KT + 1) / 4 groups of 4 taps; one
        //     v_mfma_scale_f32_16x16x128_f8f6f4 per group, block and row block sums a_lo b_hi + a_hi b_lo over the group's 64
        //     (channel, tap) pairs: K = 128 = k-group g (octet g & 1, tap pair g >> 1) x 2 slots of 16 B x [2^7 lo | 2^-4 hi] against
        //     [hi | 2^11 lo] of the activations, all e4m3, the common 2^7 undone by the A operand's block scale (2^-7);
        //     the last group's last slot is padding (zero weights; the lane re-reads the slot before it, finite by construction);
        //   * the even chunk's groups ride on pair-steps 0, 2, .. (buffer 0 is read until barrier 2), the odd chunk's on the last NG
        //     even pair-steps (buffer 1 is complete from barrier 1 on); the straddling pair-step (HS, odd) carries none;
        //   * rings: RH hi slots of 8 registers (pair-step p in slot p % RH, refilled behind its last block with pair-step p + RH, or
        //     with the NEXT super-chunk's pair-step p % RH -- no slot is shared by the last and the first pair-step, so nothing is
        //     moved), RC cross slots of 16 registers (use c in slot c % RC); every vector-memory operation of the loop is
        //     unconditional, the ages are those of F8Sched::age.
        // Per accumulator: cross terms of a group, then the pair's a_hi b_hi; groups and taps ascending.
        static_assert(W64, "S16: the 64-column window of the unrolled forms");
        constexpr int NB = 2 * NT, HS = (KT - 1) / 2, TB = KT * NB, NA = HS * NB, SB0 = (HS + 1) * NB;
#ifdef KX_DA_STAMPS  // diagnostic build: 10 ns ticks a wave spends in the loop's barriers (o[7]) and in its ring waits (o[6])
        unsigned long long tb0 = 0;
#define F8_BAR_T0 if (a.stamps) tb0 = __builtin_amdgcn_s_memrealtime();
#define F8_BAR_T1 if (a.stamps) acc_bar += __builtin_amdgcn_s_memrealtime() - tb0;
#define F8_WAIT_T1 if (a.stamps) acc_wait += __builtin_amdgcn_s_memrealtime() - tb0;
#else
#define F8_BAR_T0
#define F8_BAR_T1
#define F8_WAIT_T1
#endif
#ifndef KX_F8_RH
#define KX_F8_RH ((KT - 1) % 3 != 0 ? 3 : 4)
#endif
#ifndef KX_F8_RC
#define KX_F8_RC (NT == 6 ? 1 : 2)  // (the 192-column tile has 16 registers for one cross slot, not 32 for two)
#endif
        constexpr int RH = KX_F8_RH, RC = KX_F8_RC;
        using FS = F8Sched<KT, RH, RC, raw_ops>;
        constexpr int NG = FS::NG, NC = FS::NC;
        static_assert(RH <= KT && (KT - 1) % RH != 0 && NC % RC == 0, "f16f8: static ring slots");
        static_assert(2 * (NG - 1) <= HS && KT - 1 - 2 * (NG - 1) >= HS && (HS & 1), "f16f8: cross groups inside their buffer's phase");
        const int n_super = n_chunks >> 1;
        const int d_norm = tau16 ? dil : 0, d_str = tau16 ? XBUF - (KT - 1) * dil : 0;
        int d_x2 = tau16 ? 2 * dil : 0;  // a lane's first slot of a group: tap 4 m + 2 (lane >> 5)
        int d_xl = tau16 ? 0 : dil;      // its second slot in a chunk's LAST group: the next tap, or (padding) the same one again
        using v8i = __attribute__((ext_vector_type(8))) int;
        u32x4 hs[RH][2], xs[RC][4];
        // The ring loads are buffer loads: the image's descriptor in scalar registers, a per-lane byte offset that never changes
        // and a scalar offset per load -- no 64-bit vector address arithmetic and no address registers (the 192-column tile has
        // none to spare: with global loads it spilled, and a spill reload is a vector-memory operation that drains the ring).
        auto make_srd = [](const void* base) __attribute__((always_inline)) {
            const unsigned long long pa = (unsigned long long)base;
            u32x4 srd;
            srd[0] = __builtin_amdgcn_readfirstlane((unsigned)pa);
            srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
            srd[2] = 0xffffffffu;
            srd[3] = 0x00020000u;  // (raw buffer, as make_buf)
            return srd;
        };
        const u32x4 srd_h = make_srd(a.w16), srd_x = make_srd(a.w8x);
        const unsigned voff_h = 16u * (unsigned)(tau16 * tap_units + h16 * BM + wave * 32 + r16);
        const unsigned voff_x = 16u * (unsigned)((lane >> 4) * 256 + wave * 32 + r16);  // [g][slot][row][16 B]
        const unsigned tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ct * n_steps), tile_x8 = (unsigned)__builtin_amdgcn_readfirstlane(ct * n_chunks * NG);
        auto load_H = [&](int s0, u32x4 (&xh)[2]) __attribute__((always_inline)) {
            const unsigned so = (tile_h + (unsigned)__builtin_amdgcn_readfirstlane(s0 < n_steps - 2 ? s0 : n_steps - 2)) * (16u * tap_units);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(xh[0]) : "v"(voff_h), "s"(srd_h), "s"(so) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:256" : "=v"(xh[1]) : "v"(voff_h), "s"(srd_h), "s"(so) : "memory");
        };
        auto load_X = [&](int chunk, const int m, u32x4 (&x)[4]) __attribute__((always_inline)) {
            const unsigned so = (tile_x8 + (unsigned)__builtin_amdgcn_readfirstlane(chunk < n_chunks ? chunk : n_chunks - 1) * NG + m) * 16384u;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(x[0]) : "v"(voff_x), "s"(srd_x), "s"(so) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=v"(x[1]) : "v"(voff_x), "s"(srd_x), "s"(so) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:256" : "=v"(x[2]) : "v"(voff_x), "s"(srd_x), "s"(so) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2304" : "=v"(x[3]) : "v"(voff_x), "s"(srd_x), "s"(so) : "memory");
        };
        static_for<0, RH>([&](auto qc) __attribute__((always_inline)) { load_H(2 * decltype(qc)::value, hs[decltype(qc)::value]); });
        static_for<0, RC>([&](auto qc) __attribute__((always_inline)) {
            constexpr int c = decltype(qc)::value;
            load_X(c >= NG ? 1 : 0, c >= NG ? c - NG : c, xs[c]);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (once per tile: the ages of the loop are steady-state ages)
        int xl_it = h16 * XWp + r16;
        int dil_it = dil;
        half8 fh[2];
        v8i fx[2];
        auto load_blk_h = [&](const int p, const int nb, half8& fhx) __attribute__((always_inline)) {
            const int f0 = 2 * p, b0 = f0 >= KT ? 1 : 0, tp = f0 - b0 * KT;
            const uint4* xt = Xs + (xl_it + b0 * XBUF + tp * dil_it + (f0 == KT - 1 ? d_str : d_norm) + nb * 16);
            fhx = *reinterpret_cast<const half8*>(xt);
        };
        auto load_blk_x = [&](const int c, const int nb, v8i& fxx) __attribute__((always_inline)) {
            const int b0 = c >= NG ? 1 : 0, m = c - b0 * NG;
            const uint4* xt = Xs + (xl_it + 2 * XWp + b0 * XBUF + 4 * m * dil_it + d_x2 + nb * 16);
            const uint4 u0 = *xt, u1 = *(xt + (m == NG - 1 ? d_xl : dil_it));
            fxx = v8i{(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w, (int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
        };
#ifndef KX_F8_I0A_LONG
#define KX_F8_I0A_LONG 40
#endif
#ifndef KX_F8_I0A_MID
#define KX_F8_I0A_MID 16
#endif
#ifndef KX_F8_I0B
#define KX_F8_I0B 8
#endif
#ifndef KX_F8_I0A_SHORT
#define KX_F8_I0A_SHORT 4
#endif
#ifndef KX_F8_I0B_SHORT
#define KX_F8_I0B_SHORT 2
#endif
        constexpr int I0A = (KT >= 9 ? KX_F8_I0A_LONG : (KT >= 5 ? KX_F8_I0A_MID : KX_F8_I0A_SHORT)) * NT / 8,
                      I0B = (KT >= 5 ? KX_F8_I0B : KX_F8_I0B_SHORT) * NT / 8;
        static_assert(I0A < NA && I0B < NA, "f16f8: the transform needs blocks to ride on");
#ifndef KX_F8_G
#define KX_F8_G 2
#endif
        constexpr int G = KX_F8_G;
        static_assert(NB % G == 0, "f16f8: a scheduling region lies inside one pair-step");
        load_blk_h(0, 0, fh[0]);
        load_blk_x(0, 0, fx[0]);
        for (int sc = 0; sc < n_super; ++sc) {
            const int ch0 = 2 * sc, P0 = sc * KT;
            asm volatile("" : "+v"(xl_it), "+s"(dil_it), "+v"(d_x2), "+v"(d_xl));
            static_for<0, TB>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int p = i / NB, nb = i % NB, sl = p % RH, e = i & 1, ip = i + 1;
                constexpr int c = FS::cross_of(p), xsl = c >= 0 ? c % RC : 0;
                if constexpr (nb == 0) {
                    if constexpr (p == HS) {
                        F8_BAR_T0
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier 1
                        F8_BAR_T1
                        load_raw(ch0 + 2 < n_chunks ? ch0 + 2 : n_chunks - 1);
                        load_blk_h(p, 0, fh[e]);
                    }
                    if constexpr (p == HS + 1) {
                        F8_BAR_T0
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier 2
                        F8_BAR_T1
                    }
                    constexpr int ag_h = FS::age(0, p), ag_x = c >= 0 ? FS::age(1, c) : 63;
                    F8_BAR_T0
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ag_h < ag_x ? ag_h : ag_x) : "memory");
                    F8_WAIT_T1
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" : "+v"(hs[sl][0]), "+v"(hs[sl][1]));
                    if constexpr (c >= 0) asm volatile("" : "+v"(xs[xsl][0]), "+v"(xs[xsl][1]), "+v"(xs[xsl][2]), "+v"(xs[xsl][3]));
                }
                const half8 a0h = __builtin_bit_cast(half8, hs[sl][0]), a1h = __builtin_bit_cast(half8, hs[sl][1]);
                // the next block's fragments (not across barrier 1: the image they read is still being written)
                if constexpr (ip < TB && ip != HS * NB) {
                    load_blk_h(ip / NB, ip % NB, fh[e ^ 1]);
                    if constexpr (FS::cross_of(ip / NB) >= 0) load_blk_x(FS::cross_of(ip / NB), ip % NB, fx[e ^ 1]);
                }
                if constexpr (i % G == 0) __builtin_amdgcn_sched_barrier(0);
#ifdef KX_F8_NO_CROSS  // (diagnostic build: the cross-term MFMAs dropped, results wrong: timing only)
                if constexpr (false) {
#else
                if constexpr (c >= 0) {
#endif
                    const v8i x0 = v8i{(int)xs[xsl][0][0], (int)xs[xsl][0][1], (int)xs[xsl][0][2], (int)xs[xsl][0][3],
                                       (int)xs[xsl][1][0], (int)xs[xsl][1][1], (int)xs[xsl][1][2], (int)xs[xsl][1][3]};
                    const v8i x1 = v8i{(int)xs[xsl][2][0], (int)xs[xsl][2][1], (int)xs[xsl][2][2], (int)xs[xsl][2][3],
                                       (int)xs[xsl][3][0], (int)xs[xsl][3][1], (int)xs[xsl][3][2], (int)xs[xsl][3][3]};
                    // (A, B: e4m3; block scales 2^-7 on A, 1 on B)
                    acc16[0][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x0, fx[e], acc16[0][nb], 0, 0, 0, 0x78787878, 0, 0x7f7f7f7f);
                    acc16[1][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x1, fx[e], acc16[1][nb], 0, 0, 0, 0x78787878, 0, 0x7f7f7f7f);
                }
                acc16[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, fh[e], acc16[0][nb], 0, 0, 0);
                acc16[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, fh[e], acc16[1][nb], 0, 0, 0);
                // half-units of the transform riding on this block: phase A -> the odd chunk into buffer 1, phase B -> the next
                // even chunk into buffer 0 (in the last super-chunk: stale registers into an image nobody reads)
                constexpr bool inA = i < NA, inB = i >= SB0;
                constexpr int first = inA ? 0 : SB0, rel = inA ? i : i - SB0, I0 = inA ? I0A : I0B;
                constexpr int h0 = (inA || inB) ? FS::hu_before(NB, NA, HU, first, I0, rel) : 0;
                constexpr int h1 = (inA || inB) ? FS::hu_before(NB, NA, HU, first, I0, rel + 1) : 0;
                static_assert(h1 - h0 <= 2, "at most two half-units per block");
                uint4* Xdst = Xs + (inA ? 1 : 0) * XBUF;
#ifdef KX_DA_NO_XFORM  // (diagnostic build: the transform's vector work and LDS writes dropped, image stale: timing only)
                if constexpr (h1 > h0) keep_elem(h0);
                if constexpr (h1 > h0 + 1) keep_elem(h0 + 1);
                (void)Xdst;
#else
                if constexpr (h1 > h0) {
                    xform_a(h0 / 2, h0 & 1, 0);
                    xform_b();
                    xform_c(h0 / 2, h0 & 1, Xdst, 0);
                }
                if constexpr (h1 > h0 + 1) {
                    xform_a((h0 + 1) / 2, (h0 + 1) & 1, 0);
                    xform_b();
                    xform_c((h0 + 1) / 2, (h0 + 1) & 1, Xdst, 0);
                }
#endif
                if constexpr (i % G == G - 1) {  // the pipeline of the region: its half-units spread over its MFMAs by their duration
                    constexpr int ig = i - (G - 1);
                    constexpr bool gA = ig < NA, gB = ig >= SB0;
                    constexpr int gfirst = gA ? 0 : SB0, grel = gA ? ig : ig - SB0, gI0 = gA ? I0A : I0B;
                    constexpr int hg0 = (gA || gB) ? FS::hu_before(NB, NA, HU, gfirst, gI0, grel) : 0;
                    constexpr int nh = h1 > hg0 ? h1 - hg0 : 0;
                    constexpr int units = G * (c >= 0 ? 6 : 2);  // (16-cycle units: a scaled MFMA is two)
                    constexpr int per = nh > 0 ? (nh * (UVI + (KX_F8_HWCOS ? 0 : 7)) + units - 1) / units : 0;
#pragma unroll
                    for (int tg = 0; tg < G; ++tg) {
                        if (tg > 0) __builtin_amdgcn_sched_group_barrier(0x100, c >= 0 ? 3 : 1, 0);
                        if (c >= 0) {
#pragma unroll
                            for (int k2 = 0; k2 < 2; ++k2) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                if (per > 0) __builtin_amdgcn_sched_group_barrier(0x002, 2 * per, 0);
                            }
                        }
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (per > 0) __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (nb == NB - 1) {  // refills: the hi slot with pair-step p + RH (or the next super-chunk's p % RH) ..
                    load_H(2 * (P0 + (p + RH <= KT - 1 ? p + RH : KT + p % RH)), hs[sl]);
                    if constexpr (c >= 0) {  // .. the cross slot with use c + RC
                        constexpr int cn = c + RC, cw = cn >= NC ? cn - NC : cn;
                        load_X(ch0 + (cn >= NC ? 2 : 0) + (cw >= NG ? 1 : 0), cw >= NG ? cw - NG : cw, xs[xsl]);
                    }
                }
            });
            F8_BAR_T0
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier 3
            F8_BAR_T1
            load_raw(ch0 + 3 < n_chunks ? ch0 + 3 : n_chunks - 1);
            load_blk_h(0, 0, fh[0]);
            load_blk_x(0, 0, fx[0]);
        }
        (void)n_super;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the slots stay reserved until here: the last refills are still landing when the loop ends)
        asm volatile("" ::"v"(hs[0][0]), "v"(hs[0][1]), "v"(hs[1][0]), "v"(hs[1][1]), "v"(hs[RH - 1][0]), "v"(hs[RH - 1][1]));
        if constexpr (RH > 3) asm volatile("" ::"v"(hs[2][0]), "v"(hs[2][1]));
        asm volatile("" ::"v"(xs[0][0]), "v"(xs[0][1]), "v"(xs[0][2]), "v"(xs[0][3]));
        if constexpr (RC > 1) asm volatile("" ::"v"(xs[RC - 1][0]), "v"(xs[RC - 1][1]), "v"(xs[RC - 1][2]), "v"(xs[RC - 1][3]));
        static_assert(RH >= 2 && RH <= 4 && RC >= 1 && RC <= 2, "f16f8: ring sizes the code above spells out");
    } else if constexpr (S16) {
        // ================= S16 form: the unrolled loop on v_mfma_f32_16x16x32_f16 ================================
        // Why (profiles/r03_mfma_shape_probe.txt, r03_s16_form.txt): the chip is power-limited under this kernel, and the same FLOPs
        // issued as 16x16x32 instead of 32x32x16 hold a higher clock (-9.5 % wall time in a bare loop with the same LDS operand
        // reads; the 11-tap launches of the model: -2 .. -6 %).  The tile is 128 x 192 (and 128 x 128 on small grids): with 128
        // accumulator registers the two 16-register ring slots do not fit beside the input prefetch (the 256-column form spills
        // 33 - 42 VGPRs in the loop, and spills are vector-memory operations the hand-counted waits cannot tolerate); with 96 they do.
        // The window of the 192-column tile is 256 columns = four whole 64-column blocks, two per wave pair, no split block.
        // K = 32 is the chunk's 16 channels x TWO taps: lanes 0 .. 31 of an operand carry the pair's first step, lanes 32 .. 63
        // the second (k-groups of 8 = channel octets), so the weight image, the LDS image and the transform are those of the
        // 32x32x16 form.  KT is odd, so the flattened (chunk, tap) walk is paired over a SUPER-CHUNK of two chunks = KT pair-steps;
        // the middle one straddles the two chunks (last tap of the even chunk, first tap of the odd one).  The even chunk's
        // image lives in buffer 0, the odd one's in buffer 1:
        //   phase A   pair-steps 0 .. HS - 1 read buffer 0; the odd chunk's transform is dealt out behind them into buffer 1
        //   barrier 1 (buffer 1 complete), the input prefetch of the next even chunk is issued
        //   the straddling pair-step reads both buffers; barrier 2 (buffer 0 free)
        //   phase B   pair-steps HS + 1 .. KT - 1 read buffer 1; the next even chunk's transform goes into buffer 0
        //   barrier 3 (buffer 0 complete), the input prefetch of the next odd chunk is issued.
        // A ring: TWO slots of 16 registers (2 row blocks x hi/lo), pair-step p of a super-chunk in slot p & 1, refilled behind
        // its last block for pair-step p + 2.  KT is odd, so the next super-chunk's first pair would land in slot 1: it does,
        // and is MOVED to slot 0 (16 v_mov per super-chunk, after its wait) so that the slots stay static in the unrolled code.
        // Per accumulator the order is that of the other forms (a_lo b_hi, a_hi b_lo, a_hi b_hi; taps and chunks ascending), but
        // one instruction now sums 32 products: results are NOT bit-identical to the 32x32x16 forms (both tile widths of this
        // form are identical to each other -- batch invariance; against the oracle the error is the same 1e-6 class).
        static_assert(W64, "S16: the 64-column window of the unrolled forms");
        constexpr int NB = 2 * NT;          // 16-column blocks per pair-step
        constexpr int HS = (KT - 1) / 2;    // pair-steps wholly inside one chunk
        constexpr int TB = KT * NB;         // blocks per super-chunk (6 MFMAs each)
        constexpr int NA = HS * NB;         // blocks before the straddling pair-step (and behind it)
        constexpr int SB0 = (HS + 1) * NB;  // first block behind it
        const int n_super = n_chunks >> 1;
        const int d_norm = tau16 ? dil : 0, d_str = tau16 ? XBUF - (KT - 1) * dil : 0;
        half8 fh[2], fl[2];
        // B fragments of block nb of pair-step p: lane (column lane & 15, k-group lane >> 4) reads 16 B of the first step's tap
        // (lanes 0 .. 31) or the second step's (lanes 32 .. 63: + dil, or across to buffer 1 in the straddling pair)
        // (xl_it, dil_it: the loop-invariant lane base and dilation, made opaque once per super-chunk -- otherwise all KT x NB
        // fragment addresses are hoisted out of the loop and live in registers: 330 spilled VGPRs)
        int xl_it = h16 * XWp + r16;  // (an index, not a pointer: a pointer through asm loses its LDS address space -> flat loads)
        int dil_it = dil;
        auto load_blk = [&](const int p, const int nb, half8& fhx, half8& flx) __attribute__((always_inline)) {
            const int f0 = 2 * p, b0 = f0 >= KT ? 1 : 0, tp = f0 - b0 * KT;
            const uint4* xt = Xs + (xl_it + b0 * XBUF + tp * dil_it + (f0 == KT - 1 ? d_str : d_norm) + nb * 16);
            fhx = *reinterpret_cast<const half8*>(xt);
            flx = *reinterpret_cast<const half8*>(xt + 2 * XWp);
        };
        // Every vector-memory operation of the loop is issued unconditionally, so the age of a slot at its wait is a constant:
        // the refill of the OTHER slot (4 loads) and, where barrier 1 or 3 lies in between, one input prefetch batch.
        auto age_of = [](const int p) constexpr {
            if (p == 0) return raw_ops;                               // refilled behind pair-step KT - 2; barrier 3's prefetch since
            return 4 + ((HS == p - 1 || HS == p) ? raw_ops : 0);      // (p = 1: refilled at the super-chunk's start)
        };
        // first transform part: blocks into a phase (the input prefetch was issued at the phase's start in A, a pair-step
        // earlier in B)
#ifndef KX_S16_I0A_LONG
#define KX_S16_I0A_LONG 32
#endif
#ifndef KX_S16_I0A_MID
#define KX_S16_I0A_MID 16
#endif
#ifndef KX_S16_I0B
#define KX_S16_I0B 8
#endif
        constexpr int I0A = (KT >= 9 ? KX_S16_I0A_LONG : KX_S16_I0A_MID) * NT / 8, I0B = KX_S16_I0B * NT / 8;
        static_assert(I0A < NA && I0B < NA, "S16: the transform needs blocks to ride on");
#ifndef KX_S16_G
#define KX_S16_G 2
#endif
        constexpr int G = KX_S16_G;  // blocks per scheduling region
        load_blk(0, 0, fh[0], fl[0]);
        for (int sc = 0; sc < n_super; ++sc) {
            const int ch0 = 2 * sc, P0 = sc * KT;  // (pair-step P of the launch = steps 2 P, 2 P + 1)
            asm volatile("" : "+v"(xl_it), "+s"(dil_it));
            static_for<0, TB>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int p = i / NB, nb = i % NB, sl = p & 1, e = i & 1, ip = i + 1;
                if constexpr (nb == 0 && p == 0) {
                    wait_A2(age_of(0), a2h[1], a2l[1]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        a2h[0][q] = a2h[1][q];
                        a2l[0][q] = a2l[1][q];
                    }
                    asm volatile("" : "+v"(a2h[0][0]), "+v"(a2h[0][1]), "+v"(a2l[0][0]), "+v"(a2l[0][1]));  // (copies made HERE)
                    load_A16(2 * (P0 + 1), a2h[1], a2l[1]);
                } else if constexpr (nb == 0) {
                    if constexpr (p == HS) {
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier 1
                        load_raw(ch0 + 2 < n_chunks ? ch0 + 2 : n_chunks - 1);
                        load_blk(p, 0, fh[e], fl[e]);
                    }
                    if constexpr (p == HS + 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier 2
                    wait_A2(age_of(p), a2h[sl], a2l[sl]);
                }
                const half8 a0h = __builtin_bit_cast(half8, a2h[sl][0]), a1h = __builtin_bit_cast(half8, a2h[sl][1]);
                const half8 a0l = __builtin_bit_cast(half8, a2l[sl][0]), a1l = __builtin_bit_cast(half8, a2l[sl][1]);
                // the next block's fragments (not across barriers 1 and 3: the image they read is still being written)
                if constexpr (ip < TB && !(ip == HS * NB)) load_blk(ip / NB, ip % NB, fh[e ^ 1], fl[e ^ 1]);
                if constexpr (i % G == 0) __builtin_amdgcn_sched_barrier(0);
                acc16[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l, fh[e], acc16[0][nb], 0, 0, 0);
                acc16[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, fh[e], acc16[1][nb], 0, 0, 0);
                acc16[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, fl[e], acc16[0][nb], 0, 0, 0);
                acc16[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, fl[e], acc16[1][nb], 0, 0, 0);
                acc16[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, fh[e], acc16[0][nb], 0, 0, 0);
                acc16[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, fh[e], acc16[1][nb], 0, 0, 0);
                // half-units of the transform riding on this block: phase A -> the odd chunk into buffer 1, phase B -> the next
                // even chunk into buffer 0 (in the last super-chunk: stale registers into an image nobody reads)
                constexpr bool inA = i < NA, inB = i >= SB0;
                constexpr int rel = inA ? i : i - SB0, I0 = inA ? I0A : I0B;
                constexpr int h0 = (inA || inB) && rel > I0 ? ((rel - I0) * HU) / (NA - I0) : 0;
                constexpr int h1 = (inA || inB) && rel + 1 > I0 ? ((rel + 1 - I0) * HU) / (NA - I0) : 0;
                static_assert(h1 - h0 <= 2, "at most two half-units per block");
                uint4* Xdst = Xs + (inA ? 1 : 0) * XBUF;
                if constexpr (h1 > h0) {
                    xform_a(h0 / 2, h0 & 1, 0);
                    xform_b();
                    xform_c(h0 / 2, h0 & 1, Xdst, 0);
                }
                if constexpr (h1 > h0 + 1) {
                    xform_a((h0 + 1) / 2, (h0 + 1) & 1, 0);
                    xform_b();
                    xform_c((h0 + 1) / 2, (h0 + 1) & 1, Xdst, 0);
                }
                if constexpr (i % G == G - 1) {  // the pipeline of the region: the half-units spread over its 6 G MFMAs
                    constexpr int ig = i - (G - 1);
                    constexpr bool gA = ig < NA, gB = ig >= SB0;
                    constexpr int grel = gA ? ig : ig - SB0, gI0 = gA ? I0A : I0B;
                    constexpr int hg0 = (gA || gB) && grel > gI0 ? ((grel - gI0) * HU) / (NA - gI0) : 0;
                    constexpr int nh = h1 > hg0 ? h1 - hg0 : 0;
                    constexpr int per = nh > 0 ? (nh * 27 + 6 * G - 1) / (6 * G) : 0;
#pragma unroll
                    for (int tg = 0; tg < G; ++tg) {
                        if (tg > 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                        for (int k6 = 0; k6 < 6; ++k6) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (per > 0) __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (nb == NB - 1) {  // refill the slot with the pair-step two further on
                    if constexpr (p + 2 <= KT - 1) load_A16(2 * (P0 + p + 2), a2h[sl], a2l[sl]);
                    else if constexpr (p == KT - 2) load_A16(2 * (P0 + KT), a2h[1], a2l[1]);  // (slot 1: the next super-chunk's first pair)
                }
            });
            // (barrier 3 and what follows run after the last super-chunk too: one barrier and one prefetch batch nobody reads, for
            // a loop without a run-time condition in it)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier 3
            load_raw(ch0 + 3 < n_chunks ? ch0 + 3 : n_chunks - 1);
            load_blk(0, 0, fh[0], fl[0]);
        }
        (void)n_super;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(a2h[1][0]), "v"(a2h[1][1]), "v"(a2l[1][0]), "v"(a2l[1][1]));  // (reserved until here)
    } else if constexpr (W2) {
        // ================= W2 form: the unrolled loop with the waves as 2 x 2 ===================================
        // Why (profiles/r03_mfma_shape_probe.txt): the chip is power-limited under this kernel and the LDS operand reads are 7 % of
        // a bare MFMA loop's time; with 64 x 128 wave tiles a B fragment feeds six MFMAs instead of three (-4.8 % there), and the
        // MFMAs of a tile alternate between two accumulators instead of queueing on one.  The price is 16 A registers per ring
        // slot instead of 8 -- paid by a TWO-slot ring: a refill one step (24 MFMAs ~ 1 us) ahead is enough (the four-slot ring
        // of the form below cut to two measured 0.8 % FASTER, profiles/r03_w2_form.txt).
        // Tap t of a chunk sits in slot t & 1 and is refilled behind its last tile with tap t + 2.  KT is odd, so the next
        // chunk's tap 0 lands in slot 1: it does, and is moved to slot 0 at the chunk's start (16 v_mov, after its wait), so that
        // the slots stay static in the unrolled code.  Every vector-memory operation of the loop is unconditional (indices past
        // the end are clamped), so the age of a slot at its wait is a compile-time constant: the other slot's refill, or -- tap 0
        // -- the input prefetch batch issued at the chunk boundary.
        // Per accumulator the products are added in the order of every other form: results are bit-identical to them.
        constexpr int NW = NT / 2;          // column tiles of a wave
        constexpr int TILES = KT * NW;      // tiles per chunk, six MFMAs each (P1: two)
        constexpr int MPT = P1 ? 2 : 6;
        int cur = 0;
#ifndef KX_W2_BR
#define KX_W2_BR 2
#endif
        constexpr int BR = KX_W2_BR;  // B-fragment ring: tile i in entry i % BR, read BR - 1 tiles ahead
        half8 fh[BR], fl[BR];
        auto load_tile = [&](int t, int n, half8& fhx, half8& flx) __attribute__((always_inline)) {
            const uint4* xt = Xs + cur * XBUF + h * XWp + r + t * dil + wc2 * 128 + n * 32;
            fhx = *reinterpret_cast<const half8*>(xt);
            if constexpr (!P1) flx = *reinterpret_cast<const half8*>(xt + 2 * XWp);
        };
#ifndef KX_W2_I0_LONG
#define KX_W2_I0_LONG 5  // steps before the first transform part, k >= 9 (the input prefetch was issued at the chunk's start)
#endif
#ifndef KX_W2_I0_MID
#define KX_W2_I0_MID 2   // k = 5 .. 8 (3: +1.1 .. 1.7 % on the k = 7 launches)
#endif
#ifndef KX_W2_I0_SHORT
#define KX_W2_I0_SHORT 1  // k < 5
#endif
        constexpr int I0 = (KT >= 9 ? KX_W2_I0_LONG : (KT >= 5 ? KX_W2_I0_MID : KX_W2_I0_SHORT)) * NW;
#ifndef KX_W2_G
#define KX_W2_G 2
#endif
        constexpr int G = (P1 || KT < 5) ? 1 : KX_W2_G;  // tiles per scheduling region
        static_assert(NW % G == 0 && I0 < TILES, "W2: regions inside a step, tiles left for the transform");
        static_assert(TILES % BR == 0, "W2: a chunk's first tile always sits in ring entry 0");
        static_assert(BR - 1 <= NW, "W2: the ring's head start lies inside the first step");
        static_for<0, BR - 1>([&](auto qc) __attribute__((always_inline)) { load_tile(0, decltype(qc)::value, fh[decltype(qc)::value], fl[decltype(qc)::value]); });
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int s0 = ch * KT;
            static_for<0, TILES>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = i / NW, n = i % NW, sl = t & 1, e = i % BR, ip = i + BR - 1, e2 = ip % BR;
                if constexpr (n == 0 && t == 0) {
                    wait_A2(raw_ops, a2h[1], a2l[1]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        a2h[0][q] = a2h[1][q];
                        a2l[0][q] = a2l[1][q];
                    }
                    load_A2(s0 + 1, a2h[1], a2l[1]);
                } else if constexpr (n == 0) {
                    wait_A2(APL2, a2h[sl], a2l[sl]);
                }
                const half8 a0h = __builtin_bit_cast(half8, a2h[sl][0]), a1h = __builtin_bit_cast(half8, a2h[sl][1]);
                const half8 a0l = __builtin_bit_cast(half8, a2l[sl][0]), a1l = __builtin_bit_cast(half8, a2l[sl][1]);
                if constexpr (ip < TILES) load_tile(ip / NW, ip % NW, fh[e2], fl[e2]);
                if constexpr (i % G == 0) __builtin_amdgcn_sched_barrier(0);
                if constexpr (!P1) {
                    acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, fh[e], acc[0][n], 0, 0, 0);
                    acc[0][NW + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, fh[e], acc[0][NW + n], 0, 0, 0);
                    acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, fl[e], acc[0][n], 0, 0, 0);
                    acc[0][NW + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, fl[e], acc[0][NW + n], 0, 0, 0);
                }
                acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, fh[e], acc[0][n], 0, 0, 0);
                acc[0][NW + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, fh[e], acc[0][NW + n], 0, 0, 0);
                // half-units [h0, h1) of the next chunk's transform ride on this tile (in the last chunk: stale registers into an
                // image nobody reads)
                constexpr int h0 = i > I0 ? ((i - I0) * HUX) / (TILES - I0) : 0;
                constexpr int h1 = i + 1 > I0 ? ((i + 1 - I0) * HUX) / (TILES - I0) : 0;
                static_assert(h1 - h0 <= 4, "at most four half-units per tile");
                static_for<h0, h1>([&](auto hc) __attribute__((always_inline)) {
                    constexpr int hh = decltype(hc)::value;
                    if constexpr (PRE) {
                        pre_unit(hh, Xs + (cur ^ 1) * XBUF);
                    } else {
                        xform_a(hh / 2, hh & 1, ch + 1);
                        xform_b();
                        xform_c(hh / 2, hh & 1, Xs + (cur ^ 1) * XBUF, ch + 1);
                    }
                });
                if constexpr (i % G == G - 1) {  // the pipeline of the region: its half-units spread over its MFMAs
                    constexpr int ig = i - (G - 1);
                    constexpr int hg0 = ig > I0 ? ((ig - I0) * HUX) / (TILES - I0) : 0;
                    constexpr int nh = h1 - hg0;
                    constexpr int per = nh > 0 ? (nh * UVI + MPT * G - 1) / (MPT * G) : 0;
#pragma unroll
                    for (int tg = 0; tg < G; ++tg) {
                        if (tg > 0) __builtin_amdgcn_sched_group_barrier(0x100, P1 ? 1 : 2, 0);
#pragma unroll
                        for (int k6 = 0; k6 < MPT; ++k6) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (per > 0) __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (n == NW - 1) {  // refill the slot with the step two further on
                    if constexpr (t + 2 <= KT - 1) load_A2(s0 + t + 2, a2h[sl], a2l[sl]);
                    else if constexpr (t == KT - 2) load_A2(s0 + KT, a2h[1], a2l[1]);  // (slot 1: the next chunk's tap 0)
                }
            });
            // (the barrier and what follows run after the last chunk too: one barrier and one prefetch batch nobody reads, for a
            // loop without a run-time condition in it)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
            load_raw(ch + 2 < n_chunks ? ch + 2 : n_chunks - 1);
            static_for<0, BR - 1>([&](auto qc) __attribute__((always_inline)) { load_tile(0, decltype(qc)::value, fh[decltype(qc)::value], fl[decltype(qc)::value]); });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (P1) asm volatile("" ::"v"(a2h[1][0]), "v"(a2h[1][1]));  // (the last refill: reserved until here)
        else asm volatile("" ::"v"(a2h[1][0]), "v"(a2h[1][1]), "v"(a2l[1][0]), "v"(a2l[1][1]));
    } else if constexpr (KT > 0) {
        // ================= compile-time tap count (the resblock convs: k = 3, 7, 11) ============================
        // A chunk's KT steps are unrolled, so every ring slot and every transform part has a fixed place in the code:
        // the NEXT chunk's transform (24 half-units of ~25 vector instructions) is dealt out over the chunk's 8 KT column
        // tiles behind their MFMAs -- no branch, no second code version (two versions of the loop body made the register
        // allocator spill the accumulators where they merge; guarded blocks cost more in the rounds without a transform
        // than they saved in the others).  Tap t of a chunk uses ring slot t % 3; a slot is refilled after its last
        // use in the chunk with the next chunk's tap of the same slot number, so the ring needs no drain at a boundary.
        int cur = 0;
        // ring depths: R weight slots (tap t of a chunk uses slot t % R), BR column-tile entries (tile i uses entry i % BR and
        // is read BR - 1 tiles ahead).  Four weight slots with a two-entry fragment ring need fewer registers than three and
        // three (244 against 249 for k = 11) and measure the same (23.7 against 23.9 ms on the k = 11 launches): the cost of
        // the input prefetch that remains (-4.6 ms without it) is not weight loads queued behind it, nor the transform
        // parts waiting for it (starting them 8 instead of 5 steps into the chunk: no difference), but its traffic.
#ifndef KX_DA_RING_LONG
#define KX_DA_RING_LONG 4
#endif
        constexpr int R = KT >= 7 ? KX_DA_RING_LONG : 3, BR = R == 4 ? 2 : 3;
        int ages[4] = {age0, age1, age2, 0};
        u32x4 ahs[4] = {ah0, ah1, ah2, ah2}, als[4] = {al0, al1, al2, al2};
        if constexpr (R == 4) {
            load_A(3, ahs[3], als[3]);
            ages[0] += APL;
            ages[1] += APL;
            ages[2] += APL;
        }
        half8 fh[3], fl[3];
        auto load_tile = [&](int t, int n, half8& fhx, half8& flx) __attribute__((always_inline)) {
            const uint4* xt = Xs + cur * XBUF + h * XWp + r + t * dil + n * 32;
            fhx = *reinterpret_cast<const half8*>(xt);
            if constexpr (!P1) flx = *reinterpret_cast<const half8*>(xt + 2 * XWp);
        };
        load_tile(0, 0, fh[0], fl[0]);
        if constexpr (BR == 3) load_tile(0, 1, fh[1], fl[1]);
        constexpr int TILES = NT * KT;
        // the transform starts once the input prefetch (issued at the chunk's start) has had two steps to land
#ifndef KX_DA_I0_LONG
#define KX_DA_I0_LONG 5  // steps before the first transform part, k >= 9
#endif
#ifndef KX_DA_I0_MID
#define KX_DA_I0_MID 3   // k = 5 .. 8
#endif
#ifndef KX_DA_I0_SHORT_T8
#define KX_DA_I0_SHORT_T8 8  // k < 5: tiles (of the 256-column form; scaled by NT / 8) before the first transform part
#endif
        constexpr int I0 = KT >= 9 ? KX_DA_I0_LONG * NT : (KT >= 5 ? KX_DA_I0_MID * NT : KX_DA_I0_SHORT_T8 * NT / 8);
#ifndef KX_DA_GROUP_LONG
#define KX_DA_GROUP_LONG 4  // tiles per scheduling region, k >= 9
#endif
#ifndef KX_DA_GROUP_MID
#define KX_DA_GROUP_MID 2   // k = 5 .. 8
#endif
        constexpr int G0 = (P1 || KT < 5) ? 1 : (KT >= 9 ? KX_DA_GROUP_LONG : KX_DA_GROUP_MID);
        constexpr int G = G0 < NT ? G0 : NT;
        static_assert(NT % G == 0, "a group of tiles does not straddle a tap");
        for (int ch = 0; ch < n_chunks; ++ch) {
            const bool more = ch + 1 < n_chunks;
            static_for<0, TILES>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;  // tile of the chunk
                constexpr int t = i / NT, n = i % NT, sl = t % R;
                constexpr int e = i % BR, ip = i + BR - 1, e2 = ip % BR;
                if constexpr (n == 0) wait_A(ages[sl], ahs[sl], als[sl]);
                const half8 ah = __builtin_bit_cast(half8, ahs[sl]), al = __builtin_bit_cast(half8, als[sl]);
                if constexpr (ip < TILES) load_tile(ip / NT, ip % NT, fh[e2], fl[e2]);
                // Scheduling regions of G tiles.  Measured (tools/probes/mfma_valu_coissue.hip): with two waves per SIMD up to ~4
                // independent vector instructions per MFMA are free, 8 cost their full time -- and a half-unit dealt out as
                // 3 x 9 behind the three MFMAs of ONE tile cost exactly its own duration (stamps with and without the transform:
                // 19.8 k against 17.2 k cycles per k = 11 chunk).  So a half-unit is spread over the 3 G MFMAs of a group of tiles.
                if constexpr (i % G == 0) __builtin_amdgcn_sched_barrier(0);
                if constexpr (!P1) {
                    acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fh[e], acc[0][n], 0, 0, 0);
                    acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fl[e], acc[0][n], 0, 0, 0);
                }
                acc[0][n] = mfma_hi(ah, fh[e], acc[0][n]);
                // half-units [h0, h1) of the next chunk's transform ride on this tile (in the last chunk they run on
                // stale registers into the image nobody reads: cheaper than a second version of the loop)
                constexpr int h0 = i > I0 ? ((i - I0) * HUX) / (TILES - I0) : 0;
                constexpr int h1 = i + 1 > I0 ? ((i + 1 - I0) * HUX) / (TILES - I0) : 0;
                // (no run-time condition around them: a branch would put them in a block of their own, behind the MFMAs
                // instead of between them)
#ifdef KX_DA_NO_XFORM  // (diagnostic build: the transform's vector work and LDS writes dropped, image stale: compare CYCLES, not time)
                constexpr bool do_xform = false;
#else
                constexpr bool do_xform = true;
#endif
#ifdef KX_DA_NO_XFORM
                // (the loads stay alive and counted -- the hand-counted ring waits assume every prefetch batch -- only the vector
                // work and the LDS writes go)
                if constexpr (h1 > h0) keep_elem(h0);
                if constexpr (h1 > h0 + 1) keep_elem(h0 + 1);
#endif
                if constexpr (PRE) {
                    if constexpr (h1 > h0) pre_unit(h0, Xs + (cur ^ 1) * XBUF);
                    if constexpr (h1 > h0 + 1) pre_unit(h0 + 1, Xs + (cur ^ 1) * XBUF);
                } else {
                if constexpr (do_xform && h1 > h0) {
                    xform_a(h0 / 2, h0 & 1, ch + 1);
                    xform_b();
                    xform_c(h0 / 2, h0 & 1, Xs + (cur ^ 1) * XBUF, ch + 1);
                }
                if constexpr (do_xform && h1 > h0 + 1) {
                    xform_a((h0 + 1) / 2, (h0 + 1) & 1, ch + 1);
                    xform_b();
                    xform_c((h0 + 1) / 2, (h0 + 1) & 1, Xs + (cur ^ 1) * XBUF, ch + 1);
                }
                }
                static_assert(h1 - h0 <= 2, "at most two half-units per tile");
                if constexpr (G > 1) {
                    if constexpr (i % G == G - 1) {  // the pipeline of the whole group, written at its end
                        constexpr int ig = i - (G - 1);
                        constexpr int hg0 = ig > I0 ? ((ig - I0) * HUX) / (TILES - I0) : 0;
                        constexpr int nh = h1 - hg0;                                   // half-units in the group
                        constexpr int per = nh > 0 ? (nh * UVI + 3 * G - 1) / (3 * G) : 0;  // vector instructions per MFMA
#pragma unroll
                        for (int tg = 0; tg < G; ++tg) {
                            // (the fragment reads of the later tiles of the group stay ahead of their tile's MFMAs; opening EVERY tile
                            // with the next tile's reads measured the same -- 19.46 k against 19.49 k ticks per k = 11 chunk -- and made
                            // the 128-column k = 11 form spill)
                            if (tg > 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                            for (int k3 = 0; k3 < 3; ++k3) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                if (per > 0) __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
                            }
                        }
                    }
                } else if constexpr (P1) {  // (one MFMA per tile: the vector work follows it in one piece)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else if constexpr (h1 - h0 == 1) {
#pragma unroll
                    for (int k3 = 0; k3 < 3; ++k3) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);  // its share of the vector work
                    }
                } else if constexpr (h1 - h0 >= 2) {
#pragma unroll
                    for (int k3 = 0; k3 < 3; ++k3) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 18, 0);
                    }
                }
                if constexpr (i % G == G - 1) __builtin_amdgcn_sched_barrier(0);
                if constexpr (n == NT - 1) {
                    // refill this slot: with the tap three further on, or (its last use in the chunk) with the next
                    // chunk's tap of the same slot number
                    const int nxt = t + R < KT ? ch * KT + t + R : (ch + 1) * KT + sl;
                    if (t + R < KT || (more && sl < KT)) {
                        load_A(nxt, ahs[sl], als[sl]);
#pragma unroll
                        for (int o = 0; o < R; ++o) ages[o] = o == sl ? 0 : ages[o] + APL;
                    }
                }
            });
            if (more) {
#ifdef KX_DA_STAMPS
                unsigned long long tb0 = 0;
                if (a.stamps) tb0 = __builtin_amdgcn_s_memrealtime();
#endif
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef KX_DA_STAMPS
                if (a.stamps) acc_bar += __builtin_amdgcn_s_memrealtime() - tb0;
#endif
                cur ^= 1;
                if (ch + 2 < n_chunks && !(a.dbg & 1)) {
                    load_raw(ch + 2);
#pragma unroll
                    for (int o = 0; o < R; ++o) ages[o] += raw_ops;
                }
                load_tile(0, 0, fh[0], fl[0]);
                if constexpr (BR == 3) load_tile(0, 1, fh[1], fl[1]);
            }
        }
    } else {
    int cur = 0, ch = 0, tt = 0;
    // B fragments: a three-entry ring over the column tiles of the walk.  A tile's three MFMAs run back to back on its
    // accumulator (a_lo b_hi, a_hi b_lo, a_hi b_hi: the per-accumulator order of every other tile shape), so an entry is
    // free as soon as they are issued and the fragments of the tile two further on are read under them: 24 registers
    // instead of the 32 of a two-group double buffer, which is what makes room for the transform's temporaries.
    // 8 tiles per step and 3 entries: the entry of a step's first tile rotates 0, 2, 1 over the three steps of a round.
    half8 fh[3], fl[3];
    auto load_tile = [&](int t, int n, half8& fhx, half8& flx) __attribute__((always_inline)) {
        const uint4* xt = Xs + cur * XBUF + h * XWp + r + t * dil + n * 32;
        fhx = *reinterpret_cast<const half8*>(xt);
        if constexpr (!P1) flx = *reinterpret_cast<const half8*>(xt + 2 * XWp);
    };
    load_tile(0, 0, fh[0], fl[0]);
    load_tile(0, 1, fh[1], fl[1]);
    // one (chunk, tap) step on ring slot (ahx, alx) of age `age`; the other two slots' ages are o1, o2.
    // E0: ring entry of the step's first tile, a constant at every call site.
    auto step = [&](const int E0, u32x4& ahx, u32x4& alx, int& age, int& o1, int& o2, int sidx) __attribute__((always_inline)) {
        wait_A(age, ahx, alx);
        const half8 ah = __builtin_bit_cast(half8, ahx), al = __builtin_bit_cast(half8, alx);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int e = (E0 + n) % 3, e2 = (E0 + n + 2) % 3;
            // the fragments of the tile two further on (in the next step's tap once this step runs out of tiles; after a
            // chunk's last tap the new image is read behind the barrier below)
            if (n + 2 < NT)
                load_tile(tt, n + 2, fh[e2], fl[e2]);
            else if (tt + 1 < K)
                load_tile(tt + 1, n + 2 - NT, fh[e2], fl[e2]);
            // (scheduling barriers: without them the compiler sinks the reads to just before their use and a wave running
            // alone waits out the LDS latency every time)
            __builtin_amdgcn_sched_barrier(0);
            // (the transform stays at the chunk boundary in this form: its parts placed behind these MFMAs in blocks guarded
            // by a wave-uniform flag were measured 2 % SLOWER than the plain loop -- two taken branches per tile in the
            // rounds without a transform; the unrolled form above needs no guards)
            if constexpr (!P1) {
                acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fh[e], acc[0][n], 0, 0, 0);
                acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fl[e], acc[0][n], 0, 0, 0);
            }
            acc[0][n] = mfma_hi(ah, fh[e], acc[0][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (never past the end: the compiler takes the result of such an asm for dead and available at once, and would
        // hand the registers to the epilogue while the load is still on its way)
        if (sidx + 3 < n_steps) {
            load_A(sidx + 3, ahx, alx);
            age = 0;
            o1 += APL;
            o2 += APL;
        }
        if (++tt == K) {  // chunk boundary
            tt = 0;
            ++ch;
            if (ch < n_chunks) {
                if (!PRE && (a.dbg & 4096)) {  // (bit 4096: the loads are waited for and dropped, no transform, no LDS writes)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int c = 0; c < 8; ++c) asm volatile("" ::"v"(raw[j][c]));
                } else if (!(a.dbg & 1))
                    stage_from_raw(Xs + (cur ^ 1) * XBUF, ch);
                // one barrier per chunk: the image just written becomes readable, and every wave has finished reading
                // the other one before anybody overwrites it in the NEXT chunk's staging.  Only LDS traffic has to be
                // complete: the A-ring loads stay in flight across it.
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                cur ^= 1;
                if (ch + 1 < n_chunks && !(a.dbg & (1 | 2048))) {  // (bit 2048: the transform runs on stale registers)
                    load_raw(ch + 1);
                    age += raw_ops;
                    o1 += raw_ops;
                    o2 += raw_ops;
                }
                load_tile(0, 0, fh[(E0 + NT) % 3], fl[(E0 + NT) % 3]);
                load_tile(0, 1, fh[(E0 + NT + 1) % 3], fl[(E0 + NT + 1) % 3]);
            }
        }
    };
    for (int s3 = 0; s3 < n_steps; s3 += 3) {
        step(0, ah0, al0, age0, age1, age2, s3);
        if (s3 + 1 < n_steps) step(NT % 3, ah1, al1, age1, age2, age0, s3 + 1);
        if (s3 + 2 < n_steps) step((2 * NT) % 3, ah2, al2, age2, age0, age1, s3 + 2);
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (belt and braces: no hand-counted load is in flight past this point)
#ifdef KX_DA_STAMPS
    if (a.stamps) st2 = __builtin_amdgcn_s_memrealtime();
#endif
    if (a.dbg & 8) return;
    // the statistics scratch of the epilogue lives in the input buffers: everybody must be done reading them
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float2* stat_scr = reinterpret_cast<float2*>(smem16) + wave * (32 * 33);
    float* wide_scr = reinterpret_cast<float*>(stat_scr);  // (the wide store's 4 KiB transpose scratch, same per-wave region)
    // 128-column groups of four column tiles = the statistics groups of the other tile shapes
    if constexpr (S16) {
        // 64-column groups = the statistics slots of this form (both of its tile widths: batch invariance)
        static_for<0, NT / 2>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g2 = decltype(gc)::value;
            conv_store_group16<EPI_ROWS, NB16, 2, NT == 6>(a, acc16, 4 * g2, a.w_unscale, b, ct * BM + wave * 32, t0 + 64 * g2, lane, ncols, Lout,
                                                  tile_x * (NT / 2) + g2, stat_scr, wide_scr);
        });
    } else if constexpr (W2) {
        conv_store_group<EPI_ROWS, true>(a, *reinterpret_cast<f32x16(*)[1][4]>(&acc[0][0]), a.w_unscale, b, ct * BM + wr2 * 64, t0 + 128 * wc2, r,
                                   h, ncols, Lout, tile_x * 2 + wc2, stat_scr, wide_scr);
        conv_store_group<EPI_ROWS, true>(a, *reinterpret_cast<f32x16(*)[1][4]>(&acc[0][4]), a.w_unscale, b, ct * BM + wr2 * 64 + 32, t0 + 128 * wc2,
                                   r, h, ncols, Lout, tile_x * 2 + wc2, stat_scr, wide_scr);
    } else if constexpr (NT == 8) {
        conv_store_group<EPI_ROWS>(a, *reinterpret_cast<f32x16(*)[1][4]>(&acc[0][0]), a.w_unscale, b, ct * BM + wave * 32, t0, r, h,
                                   ncols, Lout, tile_x * 2, stat_scr, wide_scr);
        conv_store_group<EPI_ROWS>(a, *reinterpret_cast<f32x16(*)[1][4]>(&acc[0][4]), a.w_unscale, b, ct * BM + wave * 32, t0 + 128, r,
                                   h, ncols, Lout, tile_x * 2 + 1, stat_scr, wide_scr);
    } else {
        conv_store_group<EPI_ROWS>(a, acc, a.w_unscale, b, ct * BM + wave * 32, t0, r, h, ncols, Lout, tile_x, stat_scr, wide_scr);
    }
#ifdef KX_DA_STAMPS
    if (a.stamps && tid == 0) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        unsigned long long* o = a.stamps + (unsigned long long)lin * 8;
        o[0] = st0; o[1] = st1; o[2] = st2;
        __builtin_amdgcn_s_waitcnt(0);
        o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = __builtin_amdgcn_s_getreg(63492);  // HW_REG_HW_ID
        o[5] = __builtin_readcyclecounter() - cyc0;
        o[6] = F8 ? acc_wait : __builtin_amdgcn_s_getreg(63508);  // HW_REG_XCC_ID; F8 forms: 10 ns ticks in the ring waits of the main loop (wave 0)
        o[7] = acc_bar;  // 10 ns ticks spent in the chunk barriers of the main loop (wave 0)
    }
#endif
}

// The launchers below exist twice: this translation unit instantiates the f16x3 kernels (P1 = false); conv_f16x3_da_p1.hip
// defines KX_DA_P1 and includes this file for the reduced-precision ones, so that the two sets compile side by side.
#ifdef KX_DA_P1
constexpr bool DA_P1 = true;
#else
constexpr bool DA_P1 = false;
#endif

#ifdef KX_DA_W2
constexpr bool DA_W2 = true;
#else
constexpr bool DA_W2 = false;
#endif

#ifdef KX_DA_S16
constexpr bool DA_S16 = true;
#else
constexpr bool DA_S16 = false;
#endif

#ifdef KX_DA_F8
constexpr bool DA_F8 = true;
#else
constexpr bool DA_F8 = false;
#endif

template <int ACT, int KT, int NTT, bool W2X = DA_W2, bool PREX = false, bool BFX = false>
static void launch_da_inst(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    auto kern = conv1d_f16x3_da_kernel<ACT, KT, NTT, DA_P1, W2X, DA_S16, PREX, BFX, DA_F8>;
    constexpr int BN = 32 * NTT;
    // two input buffers (48 / 32 KiB), and never less than the statistics scratch of the epilogue (4 waves x 8.25 KiB)
    constexpr size_t lds_x = 16 * (size_t)2 * 4 * (BN + 128), lds_scr = 4 * 32 * 33 * sizeof(float2);
    constexpr size_t lds = lds_x > lds_scr ? lds_x : lds_scr;
    dim3 grid((max_cols + BN - 1) / BN, (a.Cout + 127) / 128, B);
    KX_REQUIRE(grid.x > 0 && grid.y > 0 && grid.y < 65536 && B > 0 && B < 65536, "conv1d f16x3 da: bad grid");
    if (a.tile_prefix) {  // flat tile list: the host counted the live column tiles for THIS tile width (a.flat_tiles_host)
        KX_REQUIRE(a.flat_ny == (int)grid.y && a.flat_B == B && a.flat_bn_host == BN && a.flat_tiles_host > 0 &&
                       (long)a.flat_tiles_host * grid.y < (1L << 30),
                   "conv1d f16x3 da: flat tile list does not match the launch");
        grid = dim3((unsigned)a.flat_tiles_host * grid.y, 1, 1);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    KX_HIP(hipGetLastError());
}

#ifdef KX_DA_PRE
// The PRE forms (conv_f16x3_da_pre.hip defines KX_DA_PRE and includes this file): the input is a pre-split image (a.x16).  Which
// launches get one is decided from the layer's shape alone (conv16_pre_shape, conv_f16x3_pre.hip), never from the batch.
void launch_conv1d_f16x3_da_pre(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn) {
    KX_REQUIRE(a.x16 != nullptr && a.x16_ld > 0 && !a.in_up2 && !a.prec1 && a.stride == 1 && a.merge_T == 0,
               "conv1d f16x3 da pre: launch not eligible");
    KX_REQUIRE(bn == 256 || bn == 128, "conv1d f16x3 da pre: tile of 256 or 128 columns");
    // the chunk term of the image offsets is a 32-bit scalar: n_chunks x four planes x x16_ld x 16 B per utterance
    KX_REQUIRE((long)a.n_chunks16 * 64 * a.x16_ld < (1L << 31), "conv1d f16x3 da pre: image of one utterance beyond 2 GiB");
    // small grids (the 128-column tile was chosen): the narrow form without staging, 32 rows x 128 columns per workgroup
    if (bn == 128 && conv16_dapn_eligible(a)) {
        launch_conv1d_f16x3_dapn(a, B, max_cols, s);
        return;
    }
    const bool w64 = a.K == 3 && (a.K - 1) * a.dil <= 64;  // the unrolled 3-tap forms (W2 on the 256-column tile); else run-time taps
    if (bn == 256) {
        if (w64) launch_da_inst<ACT_NONE, 3, 8, true, true>(a, B, max_cols, s);
        else launch_da_inst<ACT_NONE, 0, 8, false, true>(a, B, max_cols, s);
    } else {
        if (w64) launch_da_inst<ACT_NONE, 3, 4, false, true>(a, B, max_cols, s);
        else launch_da_inst<ACT_NONE, 0, 4, false, true>(a, B, max_cols, s);
    }
}
#elif defined(KX_DA_F8)
// The f16f8 forms of the S16 loop (conv_f16x3_da_f8.hip defines KX_DA_S16 and KX_DA_F8 and includes this file): the S16 shapes
// whose tap count is 3 mod 4 (11, 7 and 3: all of them), when the layer carries an 8-bit cross image (ConvArgs::w8x, CONV_F16F8).
// (the 3-tap snake convs are bound by their transform, not by the matrix pipe; on this form they still gain 6 % -- 12.8 -> 12.0 ms
// over the six shapes -- from the hardware cosine, the shorter MFMA stream and the 16 x 16 shapes' clock; KX_F8_K3=0: the 2 x 2 f16x3 form)
static bool f8_k3() {
    static const int on = getenv("KX_F8_K3") ? atoi(getenv("KX_F8_K3")) : 1;
    return on != 0;
}
bool conv16_da_f8_shape(int K, int dil) { return (K == 11 || K == 7 || (K == 3 && f8_k3())) && (K - 1) * dil <= 64; }
void launch_conv1d_f16x3_da_f8(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn) {
    KX_REQUIRE(a.w8x != nullptr && conv16_da_f8_shape(a.K, a.dil) && a.act == ACT_SNAKE && a.stride == 1 && a.merge_T == 0 && !a.prec1 &&
                   a.n_chunks16 >= 2 && (a.n_chunks16 & 1) == 0,
               "conv1d f16x3 da f8: launch not eligible");
    KX_REQUIRE(bn == 192 || bn == 128, "conv1d f16x3 da f8: tile of 192 or 128 columns");
    KX_REQUIRE((long)a.Cin * a.x_ld * 4 < (1L << 32), "conv1d f16x3 da f8: input tensor of one utterance beyond 4 GiB");
    if (a.K == 11) {
        if (bn == 192) launch_da_inst<ACT_SNAKE, 11, 6>(a, B, max_cols, s);
        else launch_da_inst<ACT_SNAKE, 11, 4>(a, B, max_cols, s);
    } else if (a.K == 7) {
        if (bn == 192) launch_da_inst<ACT_SNAKE, 7, 6>(a, B, max_cols, s);
        else launch_da_inst<ACT_SNAKE, 7, 4>(a, B, max_cols, s);
    } else {
        if (bn == 192) launch_da_inst<ACT_SNAKE, 3, 6>(a, B, max_cols, s);
        else launch_da_inst<ACT_SNAKE, 3, 4>(a, B, max_cols, s);
    }
}
#elif defined(KX_DA_S16)
// The S16 forms (conv_f16x3_da_s16.hip defines KX_DA_S16 and includes this file).
// Shapes the S16 form takes (and conv16_pick_tile gives 64-column statistics slots): snake resblock convs with 11 taps and an even
// number of 16-channel chunks, and (round 4) the un-dilated 7-tap ones.  KX_DA_S16=0 switches the form off, 2 keeps it to 11 taps.
bool conv16_da_s16_shape(int BM, int K, int dil, int stride, int act, int n_chunks16, bool merged, int pmode) {
    static const int on = getenv("KX_DA_S16") ? atoi(getenv("KX_DA_S16")) : 1;
    static const int da = getenv("KX_DA") ? atoi(getenv("KX_DA")) : 1;
    static const int st = getenv("KX_DA_STATIC") ? atoi(getenv("KX_DA_STATIC")) : 1;
    // 11 taps: always.  7 taps: the un-dilated launches only (measured: the big 7-tap launches gain 3 % on this form, the
    // dilated ones lose 3 %: profiles/r03_s16_form.txt; KX_DA_S16=2 keeps the 7-tap convs off it).  The choice depends on the
    // layer's shape alone, never on the batch, so an utterance's bits do not depend on what it is batched with.
    // pmode: 0 = f16x3, 1 = a reduced-precision launch (never this form), 2 = f16f8 (the layer carries an 8-bit cross image: every
    // 7-tap conv takes the form -- with two MFMA-equivalents per product the dilated ones gain 18 % on it instead of losing 3 %)
    const bool taps = K == 11 || (K == 7 && dil == 1 && on == 1) || (K == 7 && pmode == 2) || (K == 3 && pmode == 2 && conv16_da_f8_shape(3, dil));
    return on && da && st && BM == 128 && stride == 1 && !merged && pmode != 1 && act == ACT_SNAKE && taps && (K - 1) * dil <= 64 &&
           n_chunks16 >= 2 && (n_chunks16 & 1) == 0;
}
void launch_conv1d_f16x3_da_s16(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn) {
    KX_REQUIRE(conv16_da_s16_shape(128, a.K, a.dil, a.stride, a.act, a.n_chunks16, a.merge_T > 0, conv16_pmode(a)), "conv1d f16x3 da s16: launch not eligible");
    KX_REQUIRE(bn == 192 || bn == 128, "conv1d f16x3 da s16: tile of 192 or 128 columns");
    if (a.K == 11) {
        if (bn == 192) launch_da_inst<ACT_SNAKE, 11, 6>(a, B, max_cols, s);
        else launch_da_inst<ACT_SNAKE, 11, 4>(a, B, max_cols, s);
    } else {
        if (bn == 192) launch_da_inst<ACT_SNAKE, 7, 6>(a, B, max_cols, s);
        else launch_da_inst<ACT_SNAKE, 7, 4>(a, B, max_cols, s);
    }
}
#elif defined(KX_DA_W2)
// The W2 forms (conv_f16x3_da_w2.hip defines KX_DA_W2 and includes this file): the unrolled forms of the 256-column tile;
// launch_da_ntt<8> of the main translation unit forwards here.  Results are bit-identical to the forms they replace.
// (The reduced-precision mode keeps the 4 x 1 layout: its W2 instantiations measured 78.2 against 78.4 ms per step -- that mode is
// bound by the transform's vector work, not by the operand reads -- and are not built.)
bool conv16_da_w2_has(int act, int K) { return (act == ACT_SNAKE && (K == 3 || K == 7 || K == 11)) || (act == ACT_LEAKY && K == 3); }
void launch_conv1d_f16x3_da_w2(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    KX_REQUIRE(conv16_da_w2_has(a.act, a.K) && (a.K - 1) * a.dil <= 64 && !a.prec1, "conv1d f16x3 da w2: launch not eligible");
#ifdef KX_DA_AUDIT
    launch_da_inst<ACT_SNAKE, 11, 8>(a, B, max_cols, s);
    return;
#endif
    if (a.act == ACT_LEAKY) launch_da_inst<ACT_LEAKY, 3, 8>(a, B, max_cols, s);
    else if (a.K == 11) launch_da_inst<ACT_SNAKE, 11, 8>(a, B, max_cols, s);
    else if (a.K == 7) launch_da_inst<ACT_SNAKE, 7, 8>(a, B, max_cols, s);
    else launch_da_inst<ACT_SNAKE, 3, 8>(a, B, max_cols, s);
}
#else
bool conv16_da_w2_has(int act, int K);                                                          // conv_f16x3_da_w2.hip
void launch_conv1d_f16x3_da_w2(const ConvArgs& a, int B, int max_cols, hipStream_t s);  // conv_f16x3_da_w2.hip
void launch_conv1d_f16x3_da_s16(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn);  // conv_f16x3_da_s16.hip

template <int NTT, bool BFX = false>
static void launch_da_ntt(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
#ifdef KX_DA_AUDIT  // (tests/test_asm_audit_cpu.py: the early return keeps the build short; every instantiation is still emitted)
    if (a.act == ACT_SNAKE) launch_da_inst<ACT_SNAKE, 11, 8>(a, B, max_cols, s);
    else launch_da_inst<ACT_LEAKY, 0, 4>(a, B, max_cols, s);
    return;
#endif
    // the resblock tap counts get the unrolled form with the transform between the MFMAs (KX_DA_STATIC=0: run-time form)
    static const int st = getenv("KX_DA_STATIC") ? atoi(getenv("KX_DA_STATIC")) : 1;
    // (the unrolled forms stage a window of BN + 64 columns: (K - 1) dil <= 64, true of every resblock conv of the graph)
    const bool w64 = (a.K - 1) * a.dil <= 64;
#ifndef KX_DA_P1
    // the 256-column tile's unrolled forms: the 2 x 2 wave layout (KX_DA_W2=0: 4 x 1; bit-identical either way)
    static const int w2 = getenv("KX_DA_W2") ? atoi(getenv("KX_DA_W2")) : 1;
    if (w2 && st && w64 && NTT == 8 && conv16_da_w2_has(a.act, a.K)) {
        launch_conv1d_f16x3_da_w2(a, B, max_cols, s);
        return;
    }
#endif
    if (a.act == ACT_SNAKE) {
        if (st && w64 && a.K == 11) launch_da_inst<ACT_SNAKE, 11, NTT, DA_W2, false, BFX>(a, B, max_cols, s);
        else if (st && w64 && a.K == 7) launch_da_inst<ACT_SNAKE, 7, NTT, DA_W2, false, BFX>(a, B, max_cols, s);
        else if (st && w64 && a.K == 3) launch_da_inst<ACT_SNAKE, 3, NTT, DA_W2, false, BFX>(a, B, max_cols, s);
        else launch_da_inst<ACT_SNAKE, 0, NTT, DA_W2, false, BFX>(a, B, max_cols, s);
    } else if (a.act == ACT_LEAKY) {
        if (st && w64 && a.K == 3) launch_da_inst<ACT_LEAKY, 3, NTT, DA_W2, false, BFX>(a, B, max_cols, s);
        else launch_da_inst<ACT_LEAKY, 0, NTT, DA_W2, false, BFX>(a, B, max_cols, s);
    } else
        launch_da_inst<ACT_NONE, 0, NTT, DA_W2, false, BFX>(a, B, max_cols, s);
}

#endif  // KX_DA_W2

#if defined(KX_DA_W2) || defined(KX_DA_S16) || defined(KX_DA_PRE)
#elif defined(KX_DA_P1)
// bn: 256 or 128, as launch_conv1d_f16x3_da (which forwards here when a.prec1 is set)
void launch_conv1d_f16x3_da_p1(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn) {
    if (a.prec1 == 2) {  // bf16 operands: the bf16 form of the weight image
        KX_REQUIRE(a.w16b != nullptr, "conv1d f16x3 da p1: no bf16 weight image");
        ConvArgs b16 = a;
        b16.w16 = a.w16b;
        if (bn == 256) launch_da_ntt<8, true>(b16, B, max_cols, s);
        else launch_da_ntt<4, true>(b16, B, max_cols, s);
        return;
    }
    if (bn == 256)
        launch_da_ntt<8>(a, B, max_cols, s);
    else
        launch_da_ntt<4>(a, B, max_cols, s);
}
#else
void launch_conv1d_f16x3_da_p1(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn);  // conv_f16x3_da_p1.hip
void launch_conv1d_f16x3_da_pre(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn);  // conv_f16x3_da_pre.hip

// bn: 256 (chip-filling launches) or 128 (small grids); the statistics slots are 128 columns wide either way
void launch_conv1d_f16x3_da(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn) {
    KX_REQUIRE(conv16_da_eligible(128, a.K, a.dil, a.stride, a.merge_T > 0), "conv1d f16x3 da: launch not eligible");
    KX_REQUIRE(a.n_chunks16 == (a.Cin + CK16 - 1) / CK16 && a.w16 != nullptr, "conv1d f16x3 da: weights not packed");
    KX_REQUIRE(a.epi != EPI_GELU_NEW, "conv1d f16x3 da: no gelu epilogue");
    KX_REQUIRE(bn == 256 || bn == 128 || bn == 192, "conv1d f16x3 da: tile of 256, 192 or 128 columns");
    if (max_cols <= 0) return;
    if (a.x16) {  // pre-split input image (conv_f16x3_pre.hip decides which layers get one)
        KX_REQUIRE(bn != 192, "conv1d f16x3 da: no pre-split form of the 192-column tile");
        launch_conv1d_f16x3_da_pre(a, B, max_cols, s, bn);
        return;
    }
    // 7 / 11-tap snake convs: the 16x16x32 form on its 192- or 128-column tile (conv16_pick_tile chose bn and the 64-column
    // statistics slots for it by the same predicate)
    if (conv16_da_s16_shape(128, a.K, a.dil, a.stride, a.act, a.n_chunks16, a.merge_T > 0, conv16_pmode(a))) {
        if (a.w8x && conv16_da_f8_shape(a.K, a.dil)) launch_conv1d_f16x3_da_f8(a, B, max_cols, s, bn == 128 ? 128 : 192);  // (CONV_F16F8)
        else launch_conv1d_f16x3_da_s16(a, B, max_cols, s, bn == 128 ? 128 : 192);
        return;
    }
    KX_REQUIRE(bn != 192, "conv1d f16x3 da: the 192-column tile exists in the S16 form only");
    if (a.prec1) {
        launch_conv1d_f16x3_da_p1(a, B, max_cols, s, bn);
        return;
    }
    static const int dephase = getenv("KX_DEPHASE") ? atoi(getenv("KX_DEPHASE")) : 0;  // permille of a tile's estimated time
    static const int dephase_mode = getenv("KX_DEPHASE_MODE") ? atoi(getenv("KX_DEPHASE_MODE")) : 1;
    if (dephase > 0 && bn == 256) {  // (diagnostic, profiles/r03_lanes_dephase.txt: no effect)
        ConvArgs d = a;
        const long grid_n = (long)((max_cols + 255) / 256) * ((a.Cout + 127) / 128) * B;
        // a tile: n_chunks x K x 8 column tiles x 3 MFMAs of 32 cycles, two waves per SIMD, ~80 % pipe use; + the epilogue
        const double tile_cycles = (double)a.n_chunks16 * a.K * 8 * 96 * 2.5 + 50000.0;
        d.dephase_cycles = grid_n >= 1024 ? (int)(tile_cycles * dephase / 1000.0) : 0;
        d.dephase_mode = dephase_mode;
        launch_da_ntt<8>(d, B, max_cols, s);
        return;
    }
    if (bn == 256)
        launch_da_ntt<8>(a, B, max_cols, s);
    else
        launch_da_ntt<4>(a, B, max_cols, s);
}
#endif

}  // namespace kx
