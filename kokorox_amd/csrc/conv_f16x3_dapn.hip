// The narrow form of the pre-split direct-A conv for SMALL grids: 32 output rows x 128 columns per workgroup, no input staging.
//
// Why (profiles/r05_b1_timeline_*.txt): at batch 1 the 1024-row decoder convs are 4 column tiles x 8 row tiles = 32 workgroups of
// the 128 x 128 form on 256 CUs, each walking 69 chunks x 3 taps alone (one wave per SIMD, every barrier, LDS read and ring wait
// exposed): 110 us per launch, eight of them in a row on the forward's critical path, and the polyphase upsamplers likewise.
// With the input already a pre-split image in the B-operand layout ([hi|lo][octet][column][8 halves], conv_f16x3_pre.hip) a lane's
// B fragment of a (chunk, tap) step IS one 16-byte piece of the image: lane (column r, k-half h) loads it straight into the register
// the MFMA reads, as it does its A fragment.  No LDS, no barrier, no transform; four times as many workgroups (32-row tiles), each
// wave one 32 x 32 accumulator; an eight-slot ring of inline-asm loads with constant waits (every load unconditional: steps past
// the end re-read the last one and meet a zero B fragment).
// Per accumulator the products are added in the order of every other form (chunks and taps ascending; a_lo b_hi, a_hi b_lo,
// a_hi b_hi), and the epilogue reproduces the stored values AND the fused InstanceNorm partial sums of the 128 x 128 form bit for
// bit (see store_with_stats), so which form a launch takes may depend on the grid: batch 1 equals a member of a batch.
#include "conv_f16x3_common.h"
#include <type_traits>

namespace kx {

template <int I, int N, class F>
__device__ __forceinline__ void static_for_n(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_n<I + 1, N>(f);
    }
}

bool conv16_dapn_eligible(const ConvArgs& a) {
    static const int on = getenv("KX_DAPN") ? atoi(getenv("KX_DAPN")) : 1;
    return on && a.x16 != nullptr && a.x16_ld > 0 && !a.in_up2 && !a.prec1 && a.stride == 1 && a.merge_T == 0 && a.Cout % 32 == 0 &&
           a.K >= 1 && a.K <= 12 && a.epi == EPI_NONE && (a.store == ST_NORMAL || a.store == ST_UPSCATTER) &&
           !(a.stat_part && (a.store != ST_NORMAL || a.accum)) && (long)a.n_chunks16 * a.K < 5000;
}

#ifndef KX_DAPN_RING
#define KX_DAPN_RING 8
#endif

__global__ __launch_bounds__(256, 2) void conv1d_f16x3_dapn_kernel(const ConvArgs a) {
    constexpr int BM = 128, R = KX_DAPN_RING, tap_units = 4 * BM;
    static_assert(R >= 3 && 4 * (R - 1) <= 63, "ring depth: the wait must fit vmcnt");
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [4][32][33]: the tile's stored values (statistics only)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int tile_x = blockIdx.x, rt = blockIdx.y, b = blockIdx.z;
    if (a.tile_prefix) {  // flat list of a ragged batch: (column tile of 128, row tile of 32), row tile fastest
        const int ny = a.Cout >> 5, l = blockIdx.x;
        const int gt = l / ny;
        rt = l - gt * ny;
        int lo = 0, hi = a.flat_B;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.tile_prefix[mid] <= gt) lo = mid;
            else hi = mid;
        }
        b = __builtin_amdgcn_readfirstlane(lo);
        tile_x = gt - a.tile_prefix[b];
    }
    const int t0 = tile_x * 128;
    const int Lin = a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    const int ncols = (a.store == ST_UPSCATTER) ? (Lin + 1) : Lout;
    if (t0 >= ncols) return;
    const int K = a.K, dil = a.dil, n_steps = a.n_chunks16 * K;
    const unsigned invK = (65536u + (unsigned)K - 1u) / (unsigned)K;  // step / K for step < 5000, K <= 12: exact
    const int n_rounds = (n_steps + R - 1) / R;

    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    // A: rows [32 rt, 32 rt + 32) = rows (rt & 3) * 32 + r of weight tile rt >> 2, image [step][hi|lo][k-half][BM][8]
    const uint4* wlane = reinterpret_cast<const uint4*>(a.w16) + (long)(rt >> 2) * n_steps * tap_units + h * BM + (rt & 3) * 32 + r;
    // B: the lane's column of the image, k-half h = channel octet h; planes of x16_ld columns x 16 B: hi octet 0 | 1, lo octet 0 | 1
    const char* img = reinterpret_cast<const char*>(a.x16) + (long)b * a.x16_bs;
    const unsigned plane = 16u * (unsigned)a.x16_ld;
    const int pbase = t0 + 32 * wave + r - a.pad;
    auto col_of = [&](int s, bool& ok) __attribute__((always_inline)) {  // input position of step s for this lane (s wave-uniform)
        const int sc = s < n_steps ? s : n_steps - 1;
        const int ch = (int)(((unsigned)sc * invK) >> 16), tap = sc - ch * K;
        const int p = pbase + tap * dil;
        ok = s < n_steps && p >= 0 && p < Lin;
        return p < 0 ? 0 : (p >= Lin ? Lin - 1 : p);
    };
    u32x4 ah[R], al[R], bh[R], bl[R];
    auto load_step = [&](int s, u32x4& a_hi, u32x4& a_lo, u32x4& b_hi, u32x4& b_lo) __attribute__((always_inline)) {
        const int sc = s < n_steps ? s : n_steps - 1;
        const int ch = (int)(((unsigned)sc * invK) >> 16);
        bool ok;
        const int pc = col_of(s, ok);
        const uint4* pa = wlane + (long)sc * tap_units;
        const char* base = img + (long)ch * 4 * plane;                 // (wave-uniform: a scalar register pair)
        const unsigned v_hi = (unsigned)h * plane + 16u * (unsigned)pc;  // (< 2^31: checked at launch)
        const unsigned v_lo = v_hi + 2u * plane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_hi) : "v"(pa) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_lo) : "v"(pa + 2 * BM) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b_hi) : "v"(v_hi), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b_lo) : "v"(v_lo), "s"(base) : "memory");
    };
    // (the wait has no operands, a scheduling barrier follows, and only then are the registers handed on: conv_f16x3_da.hip)
    auto wait_step = [&](u32x4& a_hi, u32x4& a_lo, u32x4& b_hi, u32x4& b_lo) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (R - 1)) : "memory");  // the R - 1 younger slots, four loads each
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(a_hi), "+v"(a_lo), "+v"(b_hi), "+v"(b_lo));
    };

    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
    static_for_n<0, R>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        load_step(t, ah[t], al[t], bh[t], bl[t]);
    });
    for (int rd = 0; rd < n_rounds; ++rd) {
        static_for_n<0, R>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            const int s = rd * R + t;
            wait_step(ah[t], al[t], bh[t], bl[t]);
            bool ok;
            (void)col_of(s, ok);
            const u32x4 z = {0u, 0u, 0u, 0u};
            const half8 b_h = __builtin_bit_cast(half8, ok ? bh[t] : z), b_l = __builtin_bit_cast(half8, ok ? bl[t] : z);
            const half8 a_h = __builtin_bit_cast(half8, ah[t]), a_l = __builtin_bit_cast(half8, al[t]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, b_h, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_l, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_h, acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the slot is free once its MFMAs are issued (they read their operands when they issue)
            load_step(s + R, ah[t], al[t], bh[t], bl[t]);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int sl = 0; sl < R; ++sl)  // (the ring's last loads land in registers that stay reserved until here)
        asm volatile("" ::"v"(ah[sl]), "v"(al[sl]), "v"(bh[sl]), "v"(bl[sl]));

    const int row0 = rt * 32, col0 = t0 + 32 * wave;
    if (!a.stat_part) {  // plain / residual / running-sum / scatter stores: the per-wave general form
        conv_store_tile<1, 1, EPI_ROWS, false>(a, acc, a.w_unscale, b, row0, col0, r, h, ncols, Lout, tile_x, nullptr);
        return;
    }
    // ---- store + the fused InstanceNorm partial sums of the 128-column slot, exactly as the 128 x 128 form leaves them.
    // Values: fma(acc, scale, bias) (+ residual) * out_mul, as conv_store_rmw / conv_store_wide4_t.  Sums: an interior slot
    // (all 128 columns inside the utterance) takes conv_store_wide4_t's order -- lane (row 8 i + rr, columns 4 pc .. 4 pc + 3)
    // adds its four columns of the four tiles in turn, then xor-1, xor-2, half-mirror over the row's eight lanes; an edge slot
    // takes conv_store_rmw's -- per column the four tiles in turn (masked columns as zeros), then the 32 columns in order.
    {
        const bool has_res = a.resid != nullptr;
        const float* rb = has_res ? a.resid + (long)b * a.r_bs : nullptr;
        float* yb = a.y + (long)b * a.y_bs;
        const int col = col0 + r;
        const bool cok = col < ncols;
        const int cc = cok ? col : ncols - 1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = (e & 3) + 8 * (e >> 2) + 4 * h, row = row0 + rl;
            float v = __builtin_fmaf(acc[0][0][e], a.w_unscale, a.bias ? a.bias[row] : 0.f);
            if (has_res) v += rb[(long)row * a.r_ld + cc];
            v *= a.out_mul;
            if (cok) yb[(long)row * a.y_ld + col] = v;
            xs[(wave * 32 + rl) * 33 + r] = cok ? v : 0.f;
        }
    }
    __syncthreads();
    if (t0 + 128 <= ncols) {
        const int rr = lane >> 3, pc = lane & 7, rl = 8 * wave + rr;  // wave w = row group i = w of conv_store_wide4_t
        float rs = 0.f, rq = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = xs[(n * 32 + rl) * 33 + 4 * pc + c];
                rs += x;
                rq = __builtin_fmaf(x, x, rq);
            }
        rs = dpp_add<0xB1>(rs);
        rq = dpp_add<0xB1>(rq);
        rs = dpp_add<0x4E>(rs);
        rq = dpp_add<0x4E>(rq);
        rs = dpp_add<0x141>(rs);
        rq = dpp_add<0x141>(rq);
        if (pc == 0) a.stat_part[((long)b * a.Cout + row0 + rl) * a.stat_tiles + tile_x] = make_float2(rs, rq);
    } else if (wave == 0 && h == 0) {
        float2 t = make_float2(0.f, 0.f);
        for (int j = 0; j < 32; ++j) {
            float rs = 0.f, rq = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float vm = xs[(n * 32 + r) * 33 + j];
                rs += vm;
                rq = __builtin_fmaf(vm, vm, rq);
            }
            if (j == 0) t = make_float2(rs, rq);
            else {
                t.x += rs;
                t.y += rq;
            }
        }
        a.stat_part[((long)b * a.Cout + row0 + r) * a.stat_tiles + tile_x] = t;
    }
}

void launch_conv1d_f16x3_dapn(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    KX_REQUIRE(conv16_dapn_eligible(a), "conv1d f16x3 dapn: launch not eligible");
    KX_REQUIRE(a.n_chunks16 == (a.Cin + CK16 - 1) / CK16 && a.w16 != nullptr, "conv1d f16x3 dapn: weights not packed");
    KX_REQUIRE((long)a.n_chunks16 * 64 * a.x16_ld < (1L << 31), "conv1d f16x3 dapn: image of one utterance beyond 2 GiB");
    if (max_cols <= 0) return;
    dim3 grid((max_cols + 127) / 128, a.Cout / 32, B);
    KX_REQUIRE(grid.x > 0 && grid.y > 0 && grid.y < 65536 && B > 0 && B < 65536, "conv1d f16x3 dapn: bad grid");
    if (a.tile_prefix) {
        KX_REQUIRE(a.flat_ny == (a.Cout + 127) / 128 && a.flat_B == B && a.flat_bn_host == 128 && a.flat_tiles_host > 0 &&
                       (long)a.flat_tiles_host * grid.y < (1L << 30),
                   "conv1d f16x3 dapn: flat tile list does not match the launch");
        grid = dim3((unsigned)a.flat_tiles_host * grid.y, 1, 1);
    }
    hipLaunchKernelGGL(conv1d_f16x3_dapn_kernel, grid, dim3(256), 4 * 32 * 33 * sizeof(float), s, a);
    KX_HIP(hipGetLastError());
}

}  // namespace kx
