// conv1d as an implicit GEMM on the CDNA4 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// This one kernel carries every dense contraction of the Kokoro-82M graph that the
// reference executes inside `sess.run` (kokorox/src/onn/ort_koko.rs:79): the generator
// and decoder conv1d stacks (~95 % of the FLOPs), the polyphase form of
// the two ConvTranspose1d up-samplers, and — as k = 1 convs over channel-major
// activations — the ALBERT / LSTM-input / projection linears.
//
//   Y[b][co][t] = bias[co] + sum_{ci,j} W[co][ci][j] * f(X[b][ci][t*stride + j*dil - pad])
//
// f = optional AdaIN affine ((x-mean)*scale+shift per (b,ci)) followed by leaky-ReLU or
// snake, applied while the input tile is staged into LDS, so normalisation + activation
// never make their own pass over HBM.  Zero padding is applied AFTER f, as in the model.
//
// Tiling (wave64, 4 waves / workgroup):
//   GEMM rows = output channels (BM per workgroup), columns = time (BN per workgroup),
//   K = Cin*k walked in chunks of 8 input channels x all k taps.
//   LDS: W chunk [k][8][BM] (copied verbatim from the pre-packed weight image, so the
//   A-fragment read `ws[(j*8+ci)*BM + row]` is lane-contiguous = conflict-free) and the
//   transformed X chunk [8][(BN-1)*stride + (k-1)*dil + 1] (B-fragment read is
//   lane-contiguous in time for stride 1).
//   Each wave owns MT x NT accumulators of 32x32 (f32x16 each); per K-step of 2 it issues
//   MT + NT ds_read_b32 and MT*NT MFMAs (64 cycles each), so LDS traffic is far below the
//   matrix pipe's appetite and several workgroups per CU overlap staging with MFMA.
//
// Epilogue: bias, residual add, accumulate, scale / divide, optional gelu_new, and three
// store forms: channel-major, time-major (LSTM gate pre-activations), and the polyphase
// scatter of a transposed convolution (row (p,co), column q -> out[co][s*q + p - pad]).
#include "conv_epilogue.h"
#include "kx_common.h"

namespace kx {

__device__ __forceinline__ float conv_in_act(float v, int act, float slope, float al, float inv_al) {
    if (act == ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == ACT_SNAKE) {
        float s = sinf(al * v);
        return v + inv_al * (s * s);
    }
    return v;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(const ConvArgs a) {
    constexpr int MT = BM / WM / 32;
    constexpr int NT = BN / WN / 32;
    static_assert(WM * WN == 4 && MT >= 1 && NT >= 1, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, ct = blockIdx.y;
    const int t0 = blockIdx.x * BN;

    const int Lin = a.in_len.lens[b] * a.in_len.mul + a.in_len.add;
    const int Lout = a.out_len.lens[b] * a.out_len.mul + a.out_len.add;
    // GEMM columns of this utterance: outputs, or input positions + 1 for the scatter form
    const int ncols = (a.store == ST_UPSCATTER) ? (Lin + 1) : Lout;
    if (t0 >= ncols) return;

    const int K = a.K, dil = a.dil, stride = a.stride;
    const int XW = (BN - 1) * stride + (K - 1) * dil + 1;
    const int XWp = (XW + 3) & ~3;
    float* ws = smem;                      // [K][8][BM]
    float* xs = smem + K * CONV_CK * BM;  // [8][XWp]

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wchunk = K * CONV_CK * BM;  // floats per (co_tile, chunk)
    const float* wbase = a.w + (long)ct * a.n_chunks * wchunk;
    const float* xb = a.x + (long)b * a.x_bs;
    const int p0 = t0 * stride - a.pad;

    for (int ch = 0; ch < a.n_chunks; ++ch) {
        // ---- stage the weight chunk: a verbatim, fully coalesced copy -----------------
        {
            const float4* src = reinterpret_cast<const float4*>(wbase + (long)ch * wchunk);
            float4* dst = reinterpret_cast<float4*>(ws);
            for (int i = tid; i < wchunk / 4; i += 256) dst[i] = src[i];
        }
        // ---- stage + transform the input chunk: wave w takes channels w and w+4 -------
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cil = wave + half * 4;
            const int ci = ch * CONV_CK + cil;
            const bool cok = ci < a.Cin;
            float mean = 0.f, sc = 1.f, sh = 0.f, al = 1.f, inv_al = 1.f;
            const bool has_norm = a.nmean != nullptr;
            if (cok) {
                if (has_norm) {
                    mean = a.nmean[(long)b * a.n_bs + ci];
                    sc = a.nscale[(long)b * a.n_bs + ci];
                    sh = a.nshift[(long)b * a.n_bs + ci];
                }
                if (a.act == ACT_SNAKE) {
                    al = a.alpha[ci];
                    inv_al = 1.0f / al;
                }
            }
            const float* xrow = xb + (long)ci * a.x_ld;
            float* xdst = xs + cil * XWp;
            for (int u = lane; u < XW; u += 64) {
                const int p = p0 + u;
                float v = 0.f;
                if (cok && p >= 0 && p < Lin) {
                    v = xrow[a.in_up2 ? (p >> 1) : p];
                    if (has_norm) v = (v - mean) * sc + sh;
                    v = conv_in_act(v, a.act, a.slope, al, inv_al);
                }
                xdst[u] = v;
            }
        }
        __syncthreads();
        // ---- MFMA over this chunk: K-steps of 2 = (tap j, channel pair 2m+h) ----------
        const float* wl = ws + h * BM + wm * (MT * 32) + r;
        const float* xl = xs + h * XWp + (wn * (NT * 32) + r) * stride;
        for (int j = 0; j < K; ++j) {
            const float* wj = wl + j * (CONV_CK * BM);
            const float* xj = xl + j * dil;
#pragma unroll
            for (int m2 = 0; m2 < CONV_CK / 2; ++m2) {
                float av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = wj[(2 * m2) * BM + mt * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = xj[(2 * m2) * XWp + nt * 32 * stride];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue (conv_epilogue.h) -------------------------------------------------------------
    conv_store_tile<MT, NT, 4, (BM == 128 && BN == 128)>(a, acc, 1.0f, b, ct * BM + wm * (MT * 32), t0 + wn * (NT * 32), r, h, ncols, Lout,
                            blockIdx.x * WN + wn);
}

template <int BM, int BN, int WM, int WN>
static void launch_inst(const ConvArgs& a, int B, int max_cols, hipStream_t s) {
    static DynLdsLimit lds_limit;  // raise the dynamic-LDS limit only as far as a launch needs (per device)
    auto kern = conv1d_mfma_kernel<BM, BN, WM, WN>;
    const int XW = (BN - 1) * a.stride + (a.K - 1) * a.dil + 1;
    const int XWp = (XW + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)a.K * CONV_CK * BM + (size_t)CONV_CK * XWp);
    KX_REQUIRE(lds <= 160 * 1024, "conv1d: LDS tile too large for this k/stride");
    lds_limit.ensure(reinterpret_cast<const void*>(kern), lds);
    dim3 grid((max_cols + BN - 1) / BN, (a.Cout + BM - 1) / BM, B);
    KX_REQUIRE(grid.x > 0 && grid.y > 0 && grid.y < 65536 && B > 0 && B < 65536, "conv1d: bad grid");
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    KX_HIP(hipGetLastError());
}

void launch_conv1d(const ConvArgs& a, int BM, int B, int max_cols, hipStream_t s) {
    KX_REQUIRE(a.n_chunks == (a.Cin + CONV_CK - 1) / CONV_CK, "conv1d: n_chunks mismatch");
    if (max_cols <= 0) return;
    if (BM == 128)
        launch_inst<128, 128, 2, 2>(a, B, max_cols, s);
    else if (BM == 64)
        launch_inst<64, 256, 1, 4>(a, B, max_cols, s);
    else if (BM == 32)
        launch_inst<32, 256, 1, 4>(a, B, max_cols, s);
    else
        throw Error(1, "conv1d: unsupported BM");
}

// ---- weight repacking ---------------------------------------------------------------------
size_t packed_conv_floats(int rows, int Cin, int K, int BM) {
    const size_t tiles = (rows + BM - 1) / BM, chunks = (Cin + CONV_CK - 1) / CONV_CK;
    return tiles * chunks * K * CONV_CK * BM;
}

__global__ void pack_conv_kernel(PackSrc src, float* dst, int Cout, int Cin, int K, int BM, long total) {
    const int n_chunks = (Cin + CONV_CK - 1) / CONV_CK;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long q = e;
        const int m = q % BM; q /= BM;
        const int cil = q % CONV_CK; q /= CONV_CK;
        const int j = q % K; q /= K;
        const int ch = q % n_chunks; q /= n_chunks;
        const int co = (int)q * BM + m, ci = ch * CONV_CK + cil;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            int rr = co;
            const float* p = src.p[0];
            if (rr >= src.rows[0]) { rr -= src.rows[0]; p = src.p[1];
                if (rr >= src.rows[1]) { rr -= src.rows[1]; p = src.p[2]; } }
            v = p[((long)rr * Cin + ci) * K + j];
        }
        dst[e] = v;
    }
}

void launch_pack_conv(const PackSrc& src, float* dst, int Cout, int Cin, int K, int BM, hipStream_t s) {
    const long total = (long)packed_conv_floats(Cout, Cin, K, BM);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, s, src, dst, Cout, Cin, K, BM, total);
    KX_HIP(hipGetLastError());
}

// ConvTranspose1d(Cin, Cout, k = 2s, stride s, padding (k-s)/2): out[co][s*q' + p - pad] =
// sum_ci W[ci][co][p] * x[ci][q'] + W[ci][co][p+s] * x[ci][q'-1].  As a conv over q' with 2 taps and
// left padding 1: tap 0 reads x[q'-1] (weight p+s), tap 1 reads x[q'] (weight p); GEMM row = co*s + p.
__global__ void pack_convT_kernel(const float* w, float* dst, int Cin, int Cout, int s, int BM, long total) {
    const int n_chunks = (Cin + CONV_CK - 1) / CONV_CK;
    const int rows = s * Cout, k = 2 * s;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long q = e;
        const int m = q % BM; q /= BM;
        const int cil = q % CONV_CK; q /= CONV_CK;
        const int j = q % 2; q /= 2;
        const int ch = q % n_chunks; q /= n_chunks;
        const int row = (int)q * BM + m, ci = ch * CONV_CK + cil;
        float v = 0.f;
        if (row < rows && ci < Cin) {
            const int sdiv = s, co = row / sdiv, p = row % sdiv;  // GEMM row = co * s + p (phase fastest: conv_epilogue.h, ST_UPSCATTER)
            v = w[((long)ci * Cout + co) * k + (j == 0 ? p + s : p)];
        }
        dst[e] = v;
    }
}

void launch_pack_convT(const float* w, float* dst, int Cin, int Cout, int s, int BM, hipStream_t st) {
    const long total = (long)packed_conv_floats(s * Cout, Cin, 2, BM);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_convT_kernel, dim3(blocks), dim3(256), 0, st, w, dst, Cin, Cout, s, BM, total);
    KX_HIP(hipGetLastError());
}

}  // namespace kx
