// conv1d f16x3, "direct A" form of the 128 x 256 tile: the weight fragments go from global memory (L2) straight into
// the registers the MFMAs read them from; LDS holds only the transformed input window (double-buffered).
//
// Why (profiles/r02_weight_stream_ablations.txt): in conv1d_f16x3_kernel the weights reach LDS by LDS-DMA
// (global_load_lds), 24 one-KiB copy instructions per 3-tap piece and workgroup.  Removing those copies (KX_DBG bit 2)
// takes the 128 -> 128, k = 11 launches from 25.7 to 18.0 ms per step; keeping every copy INSTRUCTION but pointing them
// all at one hot KiB (bit 256) leaves 24.8 ms: the cost is the copy instructions, not their bytes.  A CU moves
// ~12 B/clk through LDS-DMA while the fragment reads keep the LDS busy (the wave-specialised kernel's producers measured
// ~350 cycles of issue per copy), two workgroups per CU need 24 GB/s of weights, and every copy a wave issues is time it
// cannot issue MFMAs; with the weight stream gone, input staging and epilogue cost 1.1 ms each instead of 6.6 / 6.9.
//
// Here wave w owns output rows [32 w, 32 w + 32) of the 128-row tile and ALL 256 columns (1 x 8 accumulator tiles, the
// same 128 registers as the 2 x 4 split), so the A fragments of a wave are its own: a_hi / a_lo of a (chunk, tap) step
// are one 16-byte global load each per lane, coalesced (the packed image [chunk][tap][hi|lo][k-half][row][8 ch] puts a
// wave's 32 rows x 16 B back to back), issued two steps ahead of their use into a three-deep register ring.  Every
// weight byte is fetched once per workgroup, as before, but through the vector-load path (64 B/clk) and with nothing
// to wait for at a barrier: the weight-piece buffers, their four barriers per chunk and the A-operand ds_reads are
// gone; what is left is ONE barrier per 16-channel chunk (the input images alternate between two LDS buffers).
// B fragments: 16 ds_read_b128 per tap and wave, in four groups of two column tiles read one group ahead.
// Per accumulator the products are added in the same order as in conv1d_f16x3_kernel (a_lo b_hi, a_hi b_lo, a_hi b_hi
// per tap, taps and chunks ascending), and the InstanceNorm partial sums cover the same 128-column groups in the same
// order, so results are bit-identical to the other tile shapes (batch invariance, tests/test_gpu_forward.py).
// Eligible launches: 128-row weight tiles, stride 1, window <= 384 columns, not the merged token-axis form.
#include "conv_f16x3_common.h"

namespace kx {

bool conv16_da_eligible(int BM, int K, int dil, int stride, int merged) {
    return BM == 128 && stride == 1 && !merged && (K - 1) * dil + 256 <= 384;
}
bool conv16_use_da(int BM, int K, int dil, int stride, int merged) {
    static const int on = getenv("KX_DA") ? atoi(getenv("KX_DA")) : 1;
    return on && conv16_da_eligible(BM, K, dil, stride, merged);
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void conv1d_f16x3_da_kernel(const ConvArgs a) {
    constexpr int BM = 128, BN = 256, NT = 8;
    constexpr int XWp = BN + 128;           // window pitch of one image row (columns)
    constexpr int XBUF = 4 * XWp;           // uint4 per input buffer: [hi|lo][octet][XWp]
    constexpr int tap_units = 4 * BM;       // uint4 per (chunk, tap) of the packed weights: [hi|lo][k-half][BM]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    uint4* Xs = reinterpret_cast<uint4*>(smem16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, ct = blockIdx.y;
    // XCD-aware column-tile order (as conv1d_f16x3_kernel: the blocks of one XCD get a contiguous range of column tiles)
    int tile_x = blockIdx.x;
    if (a.xcd_swizzle && gridDim.x >= 16) {
        const int nx = gridDim.x;
        const int off = (int)(((long)nx * (blockIdx.y + (long)gridDim.y * blockIdx.z)) & 7);
        const int cls = (blockIdx.x + off) & 7;
        int start = 0;
        for (int c = 0; c < cls; ++c) {
            const int first = (c - off) & 7;
            start += first < nx ? (nx - first + 7) >> 3 : 0;
        }
        tile_x = start + (blockIdx.x >> 3);
    }
    const int t0 = tile_x * BN;
    const int Lin = a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    const int ncols = (a.store == ST_UPSCATTER) ? (Lin + 1) : Lout;
    if (t0 >= ncols) return;

    const int K = a.K, dil = a.dil;
    const int n_chunks = a.n_chunks16;
    const int n_steps = n_chunks * K;
    const float* xb = a.x + (long)b * a.x_bs;
    const int p0 = t0 - a.pad;
    const bool has_norm = a.nmean != nullptr;
    const int up2 = a.in_up2;
    const int Lsrc = up2 ? ((Lin + 1) >> 1) : Lin;

    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;

    // ---- input chunk staging: the scheme of conv1d_f16x3_kernel (wave w: channel octet w & 1, every other 64-column
    // block of the window; raw values and per-channel parameters of the NEXT chunk prefetched across the MFMA loop)
    constexpr int NJ = XWp / 128;
    const int g = wave & 1, jb = wave >> 1;
    float raw[NJ][8];
    float praw[4];
    int xoff[NJ];
    unsigned okmask = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = p0 + lane + 64 * (jb + 2 * j);
        okmask |= (p >= 0 && p < Lin) ? (1u << j) : 0u;
        int pi = up2 ? (p >> 1) : p;
        pi = pi < 0 ? 0 : (pi >= Lsrc ? Lsrc - 1 : pi);
        xoff[j] = pi;
    }
    const int cmax_in = a.Cin - 1;
    auto load_params = [&](int ch, float (&pv)[4]) __attribute__((always_inline)) {
        const int c = ch * CK16 + g * 8 + (lane & 7);
        const int cc = c < cmax_in ? c : cmax_in;
        pv[0] = has_norm ? a.nmean[(long)b * a.n_bs + cc] : 0.f;
        pv[1] = has_norm ? a.nscale[(long)b * a.n_bs + cc] : 1.f;
        pv[2] = has_norm ? a.nshift[(long)b * a.n_bs + cc] : 0.f;
        pv[3] = (ACT == ACT_SNAKE) ? a.alpha[cc] : 1.f;  // (its reciprocal is taken when the chunk is transformed: a
                                                          // division here would wait for the load on the spot)
    };
    auto load_raw = [&](int ch) __attribute__((always_inline)) {
        load_params(ch, praw);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ci = ch * CK16 + g * 8 + c;
            const float* row = xb + (long)(ci < cmax_in ? ci : cmax_in) * a.x_ld;
#pragma unroll
            for (int j = 0; j < NJ; ++j) raw[j][c] = row[xoff[j]];
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
                const float y = in_act<ACT>(__builtin_fmaf(x8[c] - o.m[c], o.s[c], o.h[c]), a.slope, o.al[c], o.ial[c]);  // (explicit fma: see conv_epilogue.h)
                y2[q] = y * ((ch * CK16 + g * 8 + c <= cmax_in) ? keep : 0.f);
            }
            split_pair(y2[0], y2[1], hp[c2], lp[c2]);
        }
        Xb[(0 * 2 + g) * XWp + u] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        Xb[(1 * 2 + g) * XWp + u] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
    };
    auto stage_from_raw = [&](uint4* Xb, int ch) __attribute__((always_inline)) {
        Oct o;
        unpack_params(praw, o);
#pragma unroll
        for (int j = 0; j < NJ; ++j) emit8(Xb, ch, lane + 64 * (jb + 2 * j), raw[j], o, ((okmask >> j) & 1u) != 0u);
    };

    // ---- A fragments: lane (row r, k-half h) of wave w reads 16 B at [step][hi|lo][h][32 w + r].
    // The loads are inline asm with hand-counted waits.  Left to the compiler, the ring (a loop-carried set of pending
    // loads whose age differs between the paths into the loop header) gets an s_waitcnt vmcnt(0) at every loop header:
    // the youngest weight load AND the chunk's input prefetch drained once per three steps.  Vector-memory operations of
    // a wave complete in order, so "all but the N youngest are done" is exact once N is known: age[i] counts the
    // operations issued after slot i's refill (2 per refill of another slot, raw_ops per input prefetch batch); the
    // compiler's own waits for its own loads stay safe, they can only over-wait when these loads sit among theirs.
    const uint4* wlane = reinterpret_cast<const uint4*>(a.w16) + (long)ct * n_steps * tap_units + h * BM + wave * 32 + r;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    auto load_A = [&](int s, u32x4& a_hi, u32x4& a_lo) __attribute__((always_inline)) {
        const int sc = s < n_steps ? s : n_steps - 1;  // (only the three loads of the prologue can point past the end)
        const uint4* p = wlane + (long)sc * tap_units;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_hi) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_lo) : "v"(p + 2 * BM) : "memory");
    };
    // wait until at most `age` (rounded down to a value this switch knows) vector-memory operations are outstanding.
    // The compiler takes the result of a load asm for complete the moment the asm is issued, so nothing may touch the
    // fragments before this wait: the wait has no operands (a tied "+v" operand made the register allocator copy the
    // still-empty registers elsewhere BEFORE the wait: stale fragments whenever the memory system was slow), a scheduling
    // barrier follows it, and only then does an empty asm hand the fragments on as new values: any copy the allocator
    // wants is made from that point, after the data has landed.
    auto wait_A = [&](int age, u32x4& a_hi, u32x4& a_lo) __attribute__((always_inline)) {
        if (age >= 60) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
        else if (age >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (age >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if (age >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (age >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(a_hi), "+v"(a_lo));
    };
    const int raw_ops = 24 + (has_norm ? 3 : 0) + (ACT == ACT_SNAKE ? 1 : 0);  // vector loads of one load_raw()
    // Three-deep A ring with STATIC slots: step s uses slot s % 3 and refills it for step s + 3 as soon as its MFMAs are
    // issued, so a fragment is requested two whole steps before its use and nothing ever moves between registers.  The
    // (chunk, tap) walk is flattened and unrolled by three for that.
    u32x4 ah0, al0, ah1, al1, ah2, al2;
    load_A(0, ah0, al0);
    load_A(1, ah1, al1);
    load_A(2, ah2, al2);

    load_raw(0);
    stage_from_raw(Xs, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (once: the ages below start from an empty queue)
    __syncthreads();
    int age0 = 0, age1 = 0, age2 = 0;
    if (n_chunks > 1 && !(a.dbg & 1)) {
        load_raw(1);
        age0 = age1 = age2 = raw_ops;
    }

    int cur = 0, ch = 0, tt = 0;
    half8 bh[2][2], bl[2][2];
    // B fragments of column tiles (2 q, 2 q + 1) of tap t of the current image: hi at xl, lo two image rows further
    auto load_B = [&](int t, int q, half8 (&bhx)[2], half8 (&blx)[2]) __attribute__((always_inline)) {
        const uint4* xt = Xs + cur * XBUF + h * XWp + r + t * dil + q * 64;
        bhx[0] = *reinterpret_cast<const half8*>(xt);
        bhx[1] = *reinterpret_cast<const half8*>(xt + 32);
        blx[0] = *reinterpret_cast<const half8*>(xt + 2 * XWp);
        blx[1] = *reinterpret_cast<const half8*>(xt + 2 * XWp + 32);
    };
    load_B(0, 0, bh[0], bl[0]);
    // one (chunk, tap) step on ring slot (ahx, alx) of age `age`; the other two slots' ages are o1, o2
    auto step = [&](u32x4& ahx, u32x4& alx, int& age, int& o1, int& o2, int sidx) __attribute__((always_inline)) {
        wait_A(age, ahx, alx);
        const half8 ah = __builtin_bit_cast(half8, ahx), al = __builtin_bit_cast(half8, alx);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // the next group's fragments are read under this group's MFMAs (after a tap's last group: the first group of
            // the next tap; after a chunk's last tap the new image is read behind the barrier below)
            if (q < 3)
                load_B(tt, q + 1, bh[(q + 1) & 1], bl[(q + 1) & 1]);
            else if (tt + 1 < K)
                load_B(tt + 1, 0, bh[0], bl[0]);
            // (scheduling barriers: without them the compiler sinks these reads to just before their use, single-buffers
            // the fragments, and a wave running alone waits out the LDS latency in every group)
            __builtin_amdgcn_sched_barrier(0);
            const int n0 = 2 * q, n1 = 2 * q + 1;
            // (timing ablations, KX_DBG bits: 1 no input staging, 8 no epilogue)
            acc[0][n0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[q & 1][0], acc[0][n0], 0, 0, 0);
            acc[0][n1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[q & 1][1], acc[0][n1], 0, 0, 0);
            acc[0][n0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[q & 1][0], acc[0][n0], 0, 0, 0);
            acc[0][n1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[q & 1][1], acc[0][n1], 0, 0, 0);
            acc[0][n0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[q & 1][0], acc[0][n0], 0, 0, 0);
            acc[0][n1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[q & 1][1], acc[0][n1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (never past the end: the compiler takes the result of such an asm for dead and available at once, and would
        // hand the registers to the epilogue while the load is still on its way)
        if (sidx + 3 < n_steps) {
            load_A(sidx + 3, ahx, alx);
            age = 0;
            o1 += 2;
            o2 += 2;
        }
        if (++tt == K) {  // chunk boundary
            tt = 0;
            ++ch;
            if (ch < n_chunks) {
                if (!(a.dbg & 1)) stage_from_raw(Xs + (cur ^ 1) * XBUF, ch);
                // one barrier per chunk: the image just written becomes readable, and every wave has finished reading
                // the other one before anybody overwrites it in the NEXT chunk's staging.  Only LDS traffic has to be
                // complete: the A-ring loads stay in flight across it.
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                cur ^= 1;
                if (ch + 1 < n_chunks && !(a.dbg & 1)) {
                    load_raw(ch + 1);
                    age += raw_ops;
                    o1 += raw_ops;
                    o2 += raw_ops;
                }
                load_B(0, 0, bh[0], bl[0]);
            }
        }
    };
    for (int s3 = 0; s3 < n_steps; s3 += 3) {
        step(ah0, al0, age0, age1, age2, s3);
        if (s3 + 1 < n_steps) step(ah1, al1, age1, age2, age0, s3 + 1);
        if (s3 + 2 < n_steps) step(ah2, al2, age2, age0, age1, s3 + 2);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (belt and braces: no hand-counted load is in flight past this point)
    if (a.dbg & 8) return;
    // the statistics scratch of the epilogue lives in the input buffers: everybody must be done reading them
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float2* stat_scr = reinterpret_cast<float2*>(smem16) + wave * (32 * 33);
    // two halves of four column tiles = the 128-column statistics groups of the other tile shapes
    conv_store_tile<1, 4, EPI_ROWS, false>(a, *reinterpret_cast<f32x16(*)[1][4]>(&acc[0][0]), a.w_unscale, b, ct * BM + wave * 32, t0, r,
                                           h, ncols, Lout, tile_x * 2, stat_scr);
    conv_store_tile<1, 4, EPI_ROWS, false>(a, *reinterpret_cast<f32x16(*)[1][4]>(&acc[0][4]), a.w_unscale, b, ct * BM + wave * 32,
                                           t0 + 128, r, h, ncols, Lout, tile_x * 2 + 1, stat_scr);
}

template <int ACT>
static void launch_da_inst(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    auto kern = conv1d_f16x3_da_kernel<ACT>;
    constexpr size_t lds = 16 * (size_t)2 * 4 * (256 + 128);  // two input buffers: 48 KiB
    dim3 grid((max_cols + 255) / 256, (a.Cout + 127) / 128, B);
    KX_REQUIRE(grid.x > 0 && grid.y > 0 && grid.y < 65536 && B > 0 && B < 65536, "conv1d f16x3 da: bad grid");
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    KX_HIP(hipGetLastError());
}

void launch_conv1d_f16x3_da(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    KX_REQUIRE(conv16_da_eligible(128, a.K, a.dil, a.stride, a.merge_T > 0), "conv1d f16x3 da: launch not eligible");
    KX_REQUIRE(a.n_chunks16 == (a.Cin + CK16 - 1) / CK16 && a.w16 != nullptr, "conv1d f16x3 da: weights not packed");
    KX_REQUIRE(a.epi != EPI_GELU_NEW, "conv1d f16x3 da: no gelu epilogue");
    if (max_cols <= 0) return;
    if (a.act == ACT_SNAKE)
        launch_da_inst<ACT_SNAKE>(a, B, max_cols, s);
    else if (a.act == ACT_LEAKY)
        launch_da_inst<ACT_LEAKY>(a, B, max_cols, s);
    else
        launch_da_inst<ACT_NONE>(a, B, max_cols, s);
}

}  // namespace kx
