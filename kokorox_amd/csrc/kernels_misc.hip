// Non-contraction kernels of the Kokoro-82M forward (gfx950): gathers, channel layer-norm,
// instance-norm statistics, small-T attention, LSTM recurrence, duration/alignment,
// harmonic source (SineGen), forward STFT and the inverse-STFT head.  All activations are
// channel-major [B][C][ld] with per-utterance valid lengths (ragged batch, no cross-
// utterance reads).  These are the HBM-/latency-bound passes of SURVEY.md A.4; the dense
// work is in conv_mfma.hip.
#include <cmath>

#include <atomic>

#include "kx_common.h"

namespace kx {

__device__ __forceinline__ int len_of(const LenMap& m, int b) { return m.lens[b] * m.mul + m.add; }

// ---------------------------------------------------------------------------------------
__global__ void vec_add_kernel(const float* a, const float* b, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + (b ? b[i] : 0.f);
}
void launch_vec_add(const float* a, const float* b, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(vec_add_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n);
    KX_HIP(hipGetLastError());
}

__global__ void transpose_whh_kernel(const float* w, float* out) {  // [1024 rows][256 k] -> [64 k4][1024 rows][4 k]
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 1024 * 256) {
        int r = i / 256, k = i % 256;
        out[((k >> 2) * 1024 + r) * 4 + (k & 3)] = w[i];
    }
}
void launch_transpose_whh(const float* whh, float* out, hipStream_t s) {
    hipLaunchKernelGGL(transpose_whh_kernel, dim3(1024), dim3(256), 0, s, whh, out);
    KX_HIP(hipGetLastError());
}

// ---- embeddings -------------------------------------------------------------------------
// ALBERT: word[id] + token_type[0] + position[t]  (channel-major out [B][128][ld])
// Token ids come straight from the caller's device buffer on the serving entry point (kx_infer_device): an id
// outside the table must not become an out-of-bounds read.  It is clamped for the gather and recorded in a sticky
// device word (first offender: (b << 16 | t) + 1) that the host reads at the forward's one synchronisation point
// and turns into KX_ERR_INVALID.
__device__ __forceinline__ long checked_id(long id, int n_vocab, int b, int t, unsigned* bad) {
    if (id >= 0 && id < n_vocab) return id;
    if (bad) atomicCAS(bad, 0u, (((unsigned)b << 16) | (unsigned)t) + 1u);
    return id < 0 ? 0 : n_vocab - 1;
}

__global__ void albert_embed_kernel(const int64_t* ids, long ids_stride, const float* word, const float* type0,
                                    const float* pos, float* out, long bs, int ld, const int* lens, int n_vocab,
                                    unsigned* bad) {
    const int t = blockIdx.x, b = blockIdx.y, e = threadIdx.x;
    if (t >= lens[b]) return;
    const long id = checked_id(ids[b * ids_stride + t], n_vocab, b, t, e == 0 ? bad : nullptr);
    out[b * bs + (long)e * ld + t] = (word[id * 128 + e] + type0[e]) + pos[t * 128 + e];
}
void launch_albert_embed(const int64_t* ids, long ids_stride, const float* word, const float* type0,
                         const float* pos, float* out, long bs, int ld, const int* lens, int B, int Tmax,
                         int n_vocab, unsigned* bad_id, hipStream_t s) {
    hipLaunchKernelGGL(albert_embed_kernel, dim3(Tmax, B), dim3(128), 0, s, ids, ids_stride, word, type0, pos,
                       out, bs, ld, lens, n_vocab, bad_id);
    KX_HIP(hipGetLastError());
}

__global__ void embed_kernel(const int64_t* ids, long ids_stride, const float* table, int C, float* out, long bs,
                             int ld, const int* lens, int n_vocab, unsigned* bad) {
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= lens[b]) return;
    const long id = checked_id(ids[b * ids_stride + t], n_vocab, b, t, threadIdx.x == 0 ? bad : nullptr);
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[b * bs + (long)c * ld + t] = table[id * C + c];
}
void launch_embed(const int64_t* ids, long ids_stride, const float* table, int C, float* out, long bs, int ld,
                  const int* lens, int B, int Tmax, int n_vocab, unsigned* bad_id, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3(Tmax, B), dim3(256), 0, s, ids, ids_stride, table, C, out, bs, ld, lens,
                       n_vocab, bad_id);
    KX_HIP(hipGetLastError());
}

// ---- channel layer-norm -------------------------------------------------------------------
// 16 time columns x 64 channel groups per workgroup (1024 threads: a wave = 16 columns x 4 groups); reads are 64-byte
// segments along time.  Each thread keeps its CPT = C/64 channel values in registers, so the tensor is read once:
// mean (shuffle over the wave's 4 groups, LDS over the 16 waves) -> centred sum of squares (same) -> normalise + write.
// Every user is on the token axis (T <= 512): 16-column workgroups give 9 workgroups per utterance at T = 130 where the
// former 64-column ones gave 3 - at batch 1 each of the 30 layer norms of a forward ran on three CUs (22 us each on the
// critical path).  The geometry is the same for every batch size, so an utterance's bits do not depend on the batch.
template <int CPT>
__global__ __launch_bounds__(1024) void layernorm_ch_kernel(const float* x, float* y, long bs, int ld, int C,
                                                            LenMap len, float eps, int mode, const float* g,
                                                            const float* be, int g_bs, float leaky) {
    __shared__ float red[2][16][16];
    const int tx = threadIdx.x & 15, cg = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const bool wave_lead = (threadIdx.x & 63) < 16;  // lanes whose sums stand for the wave's four groups
    const int b = blockIdx.y, t = blockIdx.x * 16 + tx;
    const int L = len_of(len, b);
    const bool ok = t < L;
    const float* xb = x + b * bs + (ok ? t : 0);
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg + 64 * i;
        v[i] = (ok && c < C) ? xb[(long)c * ld] : 0.f;
        s += v[i];
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (wave_lead) red[0][wave][tx] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[0][k][tx];
    const float mean = tot / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg + 64 * i;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (wave_lead) red[1][wave][tx] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) qt += red[1][k][tx];
    const float rstd = 1.0f / sqrtf(qt / (float)C + eps);
    if (!ok) return;
    float* yb = y + b * bs + t;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg + 64 * i;
        if (c >= C) continue;
        float o = (v[i] - mean) * rstd;
        if (mode == LN_AFFINE)
            o = o * g[c] + be[c];
        else if (mode == LN_ADA)
            o = (1.0f + g[(long)b * g_bs + c]) * o + be[(long)b * g_bs + c];
        if (leaky != 0.f) o = o > 0.f ? o : o * leaky;
        yb[(long)c * ld] = o;
    }
}
void launch_layernorm_ch(const float* x, float* y, long bs, int ld, int C, LenMap len, int B, int Lmax, float eps,
                         int mode, const float* g, const float* be, int g_bs, float leaky, hipStream_t s) {
    if (Lmax <= 0) return;
    KX_REQUIRE(C <= 768, "layernorm: at most 768 channels");
    const dim3 grid((Lmax + 15) / 16, B), block(1024);
    if (C <= 128)
        hipLaunchKernelGGL(layernorm_ch_kernel<2>, grid, block, 0, s, x, y, bs, ld, C, len, eps, mode, g, be, g_bs, leaky);
    else if (C <= 512)
        hipLaunchKernelGGL(layernorm_ch_kernel<8>, grid, block, 0, s, x, y, bs, ld, C, len, eps, mode, g, be, g_bs, leaky);
    else
        hipLaunchKernelGGL(layernorm_ch_kernel<12>, grid, block, 0, s, x, y, bs, ld, C, len, eps, mode, g, be, g_bs, leaky);
    KX_HIP(hipGetLastError());
}

// ---- ALBERT self-attention, 12 heads x 64 -----------------------------------------------------
// qkv [B][2304][ld] = rows [Q | K | V], each [64 dims][time] per head.  Flash-style on the f32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 products and sums), one wave per tile of 32 query rows, keys in blocks of 32,
// no LDS and no barriers.  Both products are computed TRANSPOSED so that the query index is the accumulator's
// column (= lane):
//   S^T[key][query] = K Q^T   A = K[key][d] (lane = key, 32 coalesced dword loads per block), B = Q (32 registers)
//   O^T[d][query]  += V^T P^T A = V[d][key] (lane = d, four 16-byte loads per 32 dims), B = P^T
// The softmax of a query then runs down the lane's own 16 accumulator registers plus one exchange between the two
// half-waves; the running maximum / normaliser / rescale are per-lane scalars; and since the contraction order of
// the second product is free, key (s & 3) + 8 (s >> 2) + 4 kk is consumed at step s from half kk - exactly the key
// that accumulator register s of S^T holds there - so P^T feeds the second MFMA straight from its registers.
// The output tile stores as whole 128-byte rows of ctx[d][time].
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* qkv, long bs, int ld, float* ctx, long cbs,
                                                             int cld, const int* lens) {
    using f32x16 = __attribute__((ext_vector_type(16))) float;
    const int b = blockIdx.x, hd = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int T = lens[b];
    const int q0 = (blockIdx.z * 4 + wave) * 32;
    if (q0 >= T) return;  // (no barriers anywhere in this kernel)
    const float* Q = qkv + b * bs + (long)(hd * 64) * ld;
    const float* Kp = Q + (long)768 * ld;
    const float* V = Q + (long)1536 * ld;
    const int qi = q0 + i < T ? q0 + i : T - 1;
    float qreg[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) qreg[s] = Q[(long)(2 * s + hh) * ld + qi];
    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) o0[e] = o1[e] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    for (int k0 = 0; k0 < T; k0 += 32) {
        const int kj = k0 + i < T ? k0 + i : T - 1;
        f32x16 st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s)
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(Kp[(long)(2 * s + hh) * ld + kj], qreg[s], st, 0, 0, 0);
        // lane (query i, half hh), register e: key k0 + (e & 3) + 8 (e >> 2) + 4 hh
        float bm = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            st[e] = j < T ? st[e] * 0.125f : -INFINITY;
            bm = fmaxf(bm, st[e]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float mn = fmaxf(mrun, bm);
        const float alpha = expf(mrun - mn);  // 0 on the first block
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            st[e] = expf(st[e] - mn);  // exp(-inf) = 0 for masked keys
            psum += st[e];
        }
        psum += __shfl_xor(psum, 32);
        lrun = lrun * alpha + psum;
        mrun = mn;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o0[e] *= alpha;
            o1[e] *= alpha;
        }
        // second product: lane = head dim d' (i), half kk (hh): V[d'][k0 + 8 g + 4 kk + 0..3], g = s >> 2
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const float* vrow = V + (long)(dt * 32 + i) * ld + k0 + 4 * hh;
            float vv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v4 = *reinterpret_cast<const float4*>(vrow + 8 * g4);  // (rows are 128-byte aligned, ld >= T rounded to 32)
                const int j = k0 + 8 * g4 + 4 * hh;
                vv[4 * g4 + 0] = j + 0 < T ? v4.x : 0.f;  // the padding of a row may hold anything
                vv[4 * g4 + 1] = j + 1 < T ? v4.y : 0.f;
                vv[4 * g4 + 2] = j + 2 < T ? v4.z : 0.f;
                vv[4 * g4 + 3] = j + 3 < T ? v4.w : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (dt == 0)
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[s], st[s], o0, 0, 0, 0);
                else
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[s], st[s], o1, 0, 0, 0);
            }
        }
    }
    if (q0 + i < T) {
        const float inv = 1.0f / lrun;
        float* cb = ctx + b * cbs + (long)(hd * 64) * cld + q0 + i;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int d = (e & 3) + 8 * (e >> 2) + 4 * hh;
            cb[(long)d * cld] = o0[e] * inv;
            cb[(long)(32 + d) * cld] = o1[e] * inv;
        }
    }
}

void launch_attention(const float* qkv, long bs, int ld, float* ctx, long cbs, int cld, const int* lens, int B,
                      int Tmax, hipStream_t s) {
    KX_REQUIRE(Tmax <= 512, "attention: T > 512");
    KX_REQUIRE(ld % 32 == 0 && ld >= ((Tmax + 31) & ~31), "attention: rows must be padded to a multiple of 32 columns");
    hipLaunchKernelGGL(attention_mfma_kernel, dim3(B, 12, (Tmax + 127) / 128), dim3(256), 0, s, qkv, bs, ld, ctx, cbs,
                       cld, lens);
    KX_HIP(hipGetLastError());
}

// ---- all AdaIN / AdaLayerNorm style projections of one call ------------------------------
__global__ __launch_bounds__(256) void style_fc_kernel(const FcDesc* desc, const float* styles, float* out,
                                                       long out_bs) {
    __shared__ float sv[128];
    const FcDesc d = desc[blockIdx.x];
    const int b = blockIdx.y;
    if (threadIdx.x < 128) sv[threadIdx.x] = styles[b * 256 + d.style_off + threadIdx.x];
    __syncthreads();
    for (int o = threadIdx.x; o < d.n_out; o += 256) {
        const float4* wr = reinterpret_cast<const float4*>(d.w + (long)o * 128);
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float4 w4 = wr[k];
            acc += w4.x * sv[4 * k] + w4.y * sv[4 * k + 1] + w4.z * sv[4 * k + 2] + w4.w * sv[4 * k + 3];
        }
        out[b * out_bs + d.out_off + o] = acc + d.b[o];
    }
}
// ---- flat tile list of a ragged batch (ConvArgs::tile_prefix) ----------------------------------------------------
__global__ void tile_prefix_kernel(LenMap len, int extra, int bn, int B, int* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int b = 0; b < B; ++b) {
        out[b] = acc;
        const int cols = len.lens[b] * len.mul + len.add + extra;
        acc += cols > 0 ? (cols + bn - 1) / bn : 0;
    }
    out[B] = acc;
}

void launch_tile_prefix(LenMap len, int extra, int bn, int B, int* out, hipStream_t s) {
    hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(64), 0, s, len, extra, bn, B, out);
    KX_HIP(hipGetLastError());
}

void launch_style_fc(const FcDesc* d_desc, int n_desc, const float* styles, float* out, long out_bs, int B,
                     hipStream_t s) {
    hipLaunchKernelGGL(style_fc_kernel, dim3(n_desc, B), dim3(256), 0, s, d_desc, styles, out, out_bs);
    KX_HIP(hipGetLastError());
}

// ---- instance-norm statistics (f64 accumulation) ------------------------------------------
// raw_out (optional): [B][C][2] float2 = the row's (sum, sum of squares) as a high and a low float part, i.e. two
// "tiles" that stats_finalize_kernel adds back up in f64: later AdaINs of the SAME tensor (the three resblocks of a
// generator stage all normalise the stage input) then skip the pass over the data.
__global__ __launch_bounds__(256) void in_stats_kernel(const float* x, long bs, int ld, LenMap len, const float* gb,
                                                       long gb_bs, int C, float* mean, float* scale, float* shift,
                                                       int n_bs, float2* raw_out) {
    __shared__ double rs[4], rq[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const int L = len_of(len, b);
    const float* row = x + b * bs + (long)c * ld;
    double s = 0.0, q = 0.0;
    // 16-byte loads, two in flight per thread (rows start on 128-byte lines: ld is a multiple of 32 floats); the tail of the
    // row goes element by element
    const int L4 = L & ~3;
    const float4* row4 = reinterpret_cast<const float4*>(row);
    int t = threadIdx.x;
    for (; 4 * (t + 256) < L4; t += 512) {
        const float4 u = row4[t], w = row4[t + 256];
        const double a0 = u.x, a1 = u.y, a2 = u.z, a3 = u.w, b0 = w.x, b1 = w.y, b2 = w.z, b3 = w.w;
        s += ((a0 + a1) + (a2 + a3)) + ((b0 + b1) + (b2 + b3));
        q += ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3));
    }
    for (; 4 * t < L4; t += 256) {
        const float4 u = row4[t];
        const double a0 = u.x, a1 = u.y, a2 = u.z, a3 = u.w;
        s += (a0 + a1) + (a2 + a3);
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    for (int e = L4 + threadIdx.x; e < L; e += 256) {
        const double v = (double)row[e];
        s += v;
        q += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        q += __shfl_down(q, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        rs[wave] = s;
        rq[wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double S = (rs[0] + rs[1]) + (rs[2] + rs[3]), Q = (rq[0] + rq[1]) + (rq[2] + rq[3]);
        if (raw_out) {
            const float sh = (float)S, qh = (float)Q;
            raw_out[((long)b * C + c) * 2 + 0] = make_float2(sh, qh);
            raw_out[((long)b * C + c) * 2 + 1] = make_float2((float)(S - (double)sh), (float)(Q - (double)qh));
        }
        const double m = L > 0 ? S / L : 0.0;
        double var = L > 0 ? Q / L - m * m : 0.0;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        const float g = gb[b * gb_bs + c], be = gb[b * gb_bs + C + c];
        mean[(long)b * n_bs + c] = (float)m;
        scale[(long)b * n_bs + c] = (1.0f + g) * rstd;
        shift[(long)b * n_bs + c] = be;
    }
}
void launch_in_stats(const float* x, long bs, int ld, int C, LenMap len, int B, const float* gb, long gb_bs,
                     float* mean, float* scale, float* shift, int n_bs, float2* raw_out, hipStream_t s) {
    hipLaunchKernelGGL(in_stats_kernel, dim3(C, B), dim3(256), 0, s, x, bs, ld, len, gb, gb_bs, C, mean, scale,
                       shift, n_bs, raw_out);
    KX_HIP(hipGetLastError());
}

// fused form: the producing conv's epilogue left per-tile partial sums (conv_epilogue.h)
__global__ __launch_bounds__(64) void stats_finalize_kernel(const float2* part, int tiles, int cols_per_tile, int C,
                                                            LenMap len, const float* gb, long gb_bs, float* mean,
                                                            float* scale, float* shift, int n_bs) {
    const int c = blockIdx.x, b = blockIdx.y;
    const int L = len_of(len, b);
    const int used = ((L + cols_per_tile - 1) / cols_per_tile);  // slots written by workgroups that did not exit early
    const float2* p = part + ((long)b * C + c) * tiles;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < used && i < tiles; i += 64) {
        const float2 v = p[i];
        s += (double)v.x;
        q += (double)v.y;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        q += __shfl_down(q, o);
    }
    if (threadIdx.x == 0) {
        const double m = L > 0 ? s / L : 0.0;
        double var = L > 0 ? q / L - m * m : 0.0;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        const float g = gb[b * gb_bs + c], be = gb[b * gb_bs + C + c];
        mean[(long)b * n_bs + c] = (float)m;
        scale[(long)b * n_bs + c] = (1.0f + g) * rstd;
        shift[(long)b * n_bs + c] = be;
    }
}
void launch_stats_finalize(const float2* part, int tiles, int cols_per_tile, int C, LenMap len, int B, const float* gb,
                           long gb_bs, float* mean, float* scale, float* shift, int n_bs, hipStream_t s) {
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(C, B), dim3(64), 0, s, part, tiles, cols_per_tile, C, len, gb, gb_bs,
                       mean, scale, shift, n_bs);
    KX_HIP(hipGetLastError());
}

// ---- voice table on device: style row lookup + the reference's un-normalised mix ---------------------
// TTSKoko::mix_styles (kokorox/src/tts/koko.rs:1255-1306): single voice = copy of row `tokens_len`;
// "a.4+b.5" = sum_k row_k * (w_k * 0.1) accumulated in order, f32, no normalisation (0.4 + 0.5 = 0.9 is used
// as is).  Multiplies and adds are kept un-fused so the result equals the host mixer bit for bit.
// kinds (optional, per utterance): 0 = the style row was given explicitly (nothing to do), 1 = single voice (copy),
// 2 = mix; null = the whole batch is single (max_mix == 1) or mixes.
__global__ void style_mix_kernel(const float* table, int n_voices, const int* voice_ids, const float* weights,
                                 int max_mix, const int* rows, const int* kinds, float* styles) {
#pragma clang fp contract(off)  // (HIP's __fmul_rn / __fadd_rn are plain * and +: without this they fuse into an FMA)
    const int b = blockIdx.x, j = threadIdx.x;
    const int kind = kinds ? kinds[b] : (max_mix == 1 ? 1 : 2);
    if (kind == 0) return;
    const int row = rows[b];
    const int* v = voice_ids + (long)b * max_mix;
    const float* w = weights + (long)b * max_mix;
    float acc = 0.f;
    if (kind == 1) {
        acc = table[((long)v[0] * 511 + row) * 256 + j];
    } else {
        for (int k = 0; k < max_mix; ++k) {
            if (v[k] < 0 || v[k] >= n_voices) continue;
            const float p = w[k] * 0.1f;
            const float term = table[((long)v[k] * 511 + row) * 256 + j] * p;
            acc = acc + term;
        }
    }
    styles[(long)b * 256 + j] = acc;
}
void launch_style_mix(const float* table, int n_voices, const int* voice_ids, const float* weights, int max_mix,
                      const int* rows, const int* kinds, float* styles, int B, hipStream_t s) {
    hipLaunchKernelGGL(style_mix_kernel, dim3(B), dim3(256), 0, s, table, n_voices, voice_ids, weights, max_mix, rows,
                       kinds, styles);
    KX_HIP(hipGetLastError());
}

// ---- output packing on device ------------------------------------------------------------------------
// f32 stereo = every sample written twice (koko.rs:1239-1246); PCM16 = (s.clamp(-1,1) * 32767) as i16,
// i.e. truncation toward zero, NaN -> 0 (kokorox-websocket/src/lib.rs:701-704).
__global__ void pack_audio_kernel(const float* audio, long audio_ld, const int* frames, int format, const int* formats,
                                  void* out, long out_stride_bytes, const long* out_off) {
    const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= 600L * frames[b]) return;
    if (formats) format = formats[b];  // (per utterance: requests of one dispatched batch may differ)
    const float sv = audio[b * audio_ld + j];
    // out_off: byte offset of utterance b in a compact output (utterances back to back); else a fixed stride
    char* ob = static_cast<char*>(out) + (out_off ? out_off[b] : b * out_stride_bytes);
    if (format == 1) {
        reinterpret_cast<float2*>(ob)[j] = make_float2(sv, sv);
    } else if (format == 2) {
        const float c = fminf(fmaxf(sv, -1.0f), 1.0f);
        reinterpret_cast<short*>(ob)[j] = (short)__float2int_rz(__fmul_rn(c, 32767.0f));
    } else {
        reinterpret_cast<float*>(ob)[j] = sv;
    }
}
void launch_pack_audio(const float* audio, long audio_ld, const int* frames, int B, int Fmax, int format, void* out,
                       long out_stride_bytes, const long* out_off, hipStream_t s, const int* formats) {
    hipLaunchKernelGGL(pack_audio_kernel, dim3((600 * Fmax + 255) / 256, B), dim3(256), 0, s, audio, audio_ld, frames,
                       format, formats, out, out_stride_bytes, out_off);
    KX_HIP(hipGetLastError());
}

// ---- diagnostics: magnitude of a conv input after its AdaIN affine ---------------------------------------------
__global__ void diag_stats_kernel(const float* x, long bs, int ld, int C, LenMap len, const float* nmean,
                                  const float* nscale, const float* nshift, int n_bs, float* out3) {
    const int c = blockIdx.x, b = blockIdx.y;
    const int L = len_of(len, b);
    const float m = nmean ? nmean[(long)b * n_bs + c] : 0.f, sc = nscale ? nscale[(long)b * n_bs + c] : 1.f,
                h = nshift ? nshift[(long)b * n_bs + c] : 0.f;
    const float* row = x + b * bs + (long)c * ld;
    float amax = 0.f, sq = 0.f;
    for (int t = threadIdx.x; t < L; t += blockDim.x) {
        const float y = (row[t] - m) * sc + h;
        amax = fmaxf(amax, fabsf(y));
        sq += y * y;
    }
    for (int o = 32; o > 0; o >>= 1) {
        amax = fmaxf(amax, __shfl_down(amax, o));
        sq += __shfl_down(sq, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(reinterpret_cast<unsigned*>(out3), __float_as_uint(amax));  // non-negative floats order as uints
        atomicAdd(out3 + 1, sq);
        if (threadIdx.x == 0) atomicAdd(out3 + 2, (float)L);
    }
}
void launch_diag_stats(const float* x, long bs, int ld, int C, LenMap len, int B, int Lmax, const float* nmean,
                       const float* nscale, const float* nshift, int n_bs, float* out3, hipStream_t s) {
    (void)Lmax;
    hipLaunchKernelGGL(diag_stats_kernel, dim3(C, B), dim3(256), 0, s, x, bs, ld, C, len, nmean, nscale, nshift, n_bs, out3);
    KX_HIP(hipGetLastError());
}

// ---- row fills / copies ---------------------------------------------------------------------
__global__ void fill_style_rows_kernel(float* dst, long bs, int ld, int row0, const float* styles, int style_off,
                                       const int* lens) {
    const int k = blockIdx.x, b = blockIdx.y;
    const float v = styles[b * 256 + style_off + k];
    const int L = lens[b];
    float* row = dst + b * bs + (long)(row0 + k) * ld;
    for (int t = threadIdx.x; t < L; t += blockDim.x) row[t] = v;
}
void launch_fill_style_rows(float* dst, long bs, int ld, int row0, const float* styles, int style_off,
                            const int* lens, int B, int Tmax, hipStream_t s) {
    (void)Tmax;
    hipLaunchKernelGGL(fill_style_rows_kernel, dim3(128, B), dim3(256), 0, s, dst, bs, ld, row0, styles, style_off,
                       lens);
    KX_HIP(hipGetLastError());
}

__global__ void copy_rows_kernel(const float* src, long sbs, int sld, float* dst, long dbs, int dld, LenMap len) {
    const int r = blockIdx.x, b = blockIdx.y;
    const int L = len_of(len, b);
    const float* s = src + b * sbs + (long)r * sld;
    float* d = dst + b * dbs + (long)r * dld;
    for (int t = threadIdx.x; t < L; t += blockDim.x) d[t] = s[t];
}
void launch_copy_rows(const float* src, long sbs, int sld, float* dst, long dbs, int dld, int rows, LenMap len,
                      int B, int Lmax, hipStream_t s) {
    (void)Lmax;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(rows, B), dim3(256), 0, s, src, sbs, sld, dst, dbs, dld, len);
    KX_HIP(hipGetLastError());
}

// ---- bidirectional LSTM recurrence (hidden 256) -------------------------------------------
// gx [B][L][2048] holds W_ih x + b_ih + b_hh for both directions (conv kernel, time-major store).  One workgroup of
// 1024 threads per (utterance, direction): thread r owns gate row r and walks W_hh^T[k][r] against h in LDS, then
// the first 256 threads apply the cell update.  A step is bound by how fast one CU can pull the 1 MB of W_hh^T
// through its L1 (~60 B/clk); the first LSTM_LDS_K of the 256 k-rows therefore stay in LDS for the whole sequence
// (36 x 4 KB = 144 KB) and the next LSTM_REG_K in registers, the rest streams from L2 every step.  The image is
// [k/4][row][4 k], so every thread moves 16-byte pieces (a quarter of the load instructions of a [k][row] image:
// the step got 20 % shorter; with 4-byte loads, more than 32 register rows made it 35-50 % longer).
constexpr int LSTM_LDS_K = 36;
constexpr int LSTM_REG_K = 64;
__global__ __launch_bounds__(1024) void lstm_kernel(const float* gx, long gx_bs, int gx_ld, const float* whhT,
                                                    float* y, long y_bs, int y_ld, LenMap len) {
    extern __shared__ __attribute__((aligned(16))) float lstm_smem[];
    float* hs = lstm_smem;                                          // [256]
    float* gates = hs + 256;                                        // [1024]
    float4* wl = reinterpret_cast<float4*>(gates + 1024);           // [LSTM_LDS_K / 4][1024] x 4 k
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int L = len_of(len, b);
    // image [k4][row][4 k]: thread = row, every access is one 16-byte piece, coalesced across the workgroup
    const float4* W4 = reinterpret_cast<const float4*>(whhT + (long)dir * 256 * 1024) + tid;
    float c = 0.f;
    if (tid < 256) hs[tid] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < LSTM_LDS_K / 4; ++k4) wl[k4 * 1024 + tid] = W4[(long)k4 * 1024];
    // ... and the next LSTM_REG_K k-rows in registers
    float4 wreg[LSTM_REG_K / 4];
#pragma unroll
    for (int k4 = 0; k4 < LSTM_REG_K / 4; ++k4) wreg[k4] = W4[(long)(LSTM_LDS_K / 4 + k4) * 1024];
    __syncthreads();
    for (int step = 0; step < L; ++step) {
        const int t = dir ? (L - 1 - step) : step;
        float acc = gx[b * gx_bs + (long)t * gx_ld + dir * 1024 + tid];
        const float4* h4 = reinterpret_cast<const float4*>(hs);
        // streamed part first in program order: its loads are in flight while the resident parts are summed
        float acc2 = 0.f;
#pragma unroll 4
        for (int k4 = (LSTM_LDS_K + LSTM_REG_K) / 4; k4 < 64; ++k4) {
            const float4 hv = h4[k4];
            const float4 w = W4[(long)k4 * 1024];
            acc2 = fmaf(w.x, hv.x, acc2);
            acc2 = fmaf(w.y, hv.y, acc2);
            acc2 = fmaf(w.z, hv.z, acc2);
            acc2 = fmaf(w.w, hv.w, acc2);
        }
#pragma unroll
        for (int k4 = 0; k4 < LSTM_LDS_K / 4; ++k4) {
            const float4 hv = h4[k4];
            const float4 w = wl[k4 * 1024 + tid];
            acc = fmaf(w.x, hv.x, acc);
            acc = fmaf(w.y, hv.y, acc);
            acc = fmaf(w.z, hv.z, acc);
            acc = fmaf(w.w, hv.w, acc);
        }
#pragma unroll
        for (int k4 = 0; k4 < LSTM_REG_K / 4; ++k4) {
            const float4 hv = h4[LSTM_LDS_K / 4 + k4];
            acc = fmaf(wreg[k4].x, hv.x, acc);
            acc = fmaf(wreg[k4].y, hv.y, acc);
            acc = fmaf(wreg[k4].z, hv.z, acc);
            acc = fmaf(wreg[k4].w, hv.w, acc);
        }
        gates[tid] = acc + acc2;
        __syncthreads();
        if (tid < 256) {
            const float ig = 1.0f / (1.0f + expf(-gates[tid]));
            const float fg = 1.0f / (1.0f + expf(-gates[256 + tid]));
            const float gg = tanhf(gates[512 + tid]);
            const float og = 1.0f / (1.0f + expf(-gates[768 + tid]));
            c = fg * c + ig * gg;
            const float hn = og * tanhf(c);
            hs[tid] = hn;
            y[b * y_bs + (long)(dir * 256 + tid) * y_ld + t] = hn;
        }
        __syncthreads();
    }
}
// One unit's cell update from its four gate pre-activations (PyTorch order i, f, g, o); returns h.  ONE definition for every
// recurrence kernel, with the multiply-add spelled out: left to -ffp-contract the same expression was fused in one kernel and
// not in another, and the resident-weights forms and their streaming fall-back must agree bit for bit.
__device__ __forceinline__ float lstm_cell(float pi, float pf, float pg, float po, float& c) {
    const float ig = 1.0f / (1.0f + expf(-pi));
    const float fg = 1.0f / (1.0f + expf(-pf));
    const float gg = tanhf(pg);
    const float og = 1.0f / (1.0f + expf(-po));
    const float t = ig * gg;
    c = __builtin_fmaf(fg, c, t);
    return og * tanhf(c);
}

// ---- the same recurrence with W_hh fully resident: two CUs per (utterance, direction) -------------------------------
// One CU cannot hold the 1 MiB of W_hh (512 KiB of registers + 160 KiB of LDS), which is why lstm_kernel streams 60 % of
// it from L2 on every step (5.5 us per step, bound by the CU's L1 fill rate).  Here the recurrence of one (utterance,
// direction) runs on TWO workgroups = two CUs.  Half hf owns hidden units [128 hf, 128 hf + 128) with all four of their
// gate rows (512 rows x 256 k = 512 KiB: 96 registers per thread + 128 KiB of LDS), so it updates its own cells locally
// and the only traffic per step is the exchange of the 128 new h values each way: 8-byte {h, tag} granules written with
// one agent-scope (sc1) store each and polled with agent-scope loads - the data-tagged hand-off of
// MI355X_MICROARCH.md ("R2 granule": a granule is valid when its tag is the expected step; no fence, no flag).
// The tag carries a per-launch epoch, so the exchange buffer never needs clearing; two slots by step parity (a half
// can only be one step ahead of its partner).  The two halves are blocks i and i + 8 of a group of 16 (same XCD);
// blocks are dispatched in order, so at most eight pairs per launch ever wait for a partner that is not resident yet.  Every poll is bounded: if the
// partner never shows up the half raises the model's sticky error word and leaves (an error, never a hang).
// Round 5: the same recurrence on FOUR workgroups (NQ = 4) for batches whose pairs leave CUs idle.  A step of the two-CU form is
// bound by the CU's LDS return path (1024 lanes x 24 16-byte reads: 16 of h, 8 of the weight tail that does not fit the 128
// registers a 1024-thread workgroup leaves a lane; packed fmas instead of scalar ones changed nothing: profiles/
// r05_experiments_not_kept.txt).  A quarter owns 64 hidden units = 256 gate rows on 512 threads, which may hold 256 registers
// each: ALL of a lane's 128 weights live in registers, the LDS traffic of a step falls to the 16 reads of h of half as many
// lanes, and four CUs share the step.  A lane's arithmetic is that of the two-CU form, operation for operation (two rows x a K
// quarter, the same order of sums): both forms give the same bits, and which one runs may depend on the batch (launch_lstm).
// Parts p of a (utterance, direction) are blocks i + 8 p of a group of 8 NQ blocks (one XCD); blocks are dispatched in order.
template <int NQ>
struct LstmParts {
    static constexpr int HU = 256 / NQ;            // hidden units of a part
    static constexpr int ROWS = 4 * HU;            // its gate rows
    static constexpr int THREADS = 2 * ROWS;       // lanes: (row pair, K quarter)
    static constexpr int REG = NQ == 2 ? 24 : 32;  // float4 (4 k) pieces of a lane's 32 held in registers, the rest in LDS
    static constexpr int LDSP = 32 - REG;
};
constexpr int LSTMP_HP = 68;            // pitch of a K quarter of h in LDS (64 values + 4: bank offset between the quarters)
template <int NQ>
__global__ __launch_bounds__(LstmParts<NQ>::THREADS) void lstm_pair_kernel(const float* gx, long gx_bs, int gx_ld, const float* whhT,
                                                         float* y, long y_bs, int y_ld, LenMap len,
                                                         unsigned long long* xchg, unsigned epoch, unsigned* err,
                                                         int n_pairs, int spin_limit, int drop_half) {
    using P = LstmParts<NQ>;
    constexpr int HU = P::HU, ROWS = P::ROWS, NT = P::THREADS, LSTMP_REG = P::REG, LSTMP_LDS = P::LDSP;
    extern __shared__ __attribute__((aligned(16))) float lstm_smem[];
    // h of the previous step (all parts) in four K quarters of 64 with a pitch of 68 floats: the four lanes of a quad read
    // four different quarters in one instruction, and 272 bytes apart they sit in different banks
    float* hs = lstm_smem;                                   // [4][LSTMP_HP]
    float* gates = hs + 4 * LSTMP_HP;                        // [ROWS] gate pre-activations of this part
    int* abort_flag = reinterpret_cast<int*>(gates + ROWS);  // [4] (one word used): a poll timed out
    float4* wl = reinterpret_cast<float4*>(gates + ROWS + 4); // [LSTMP_LDS][NT] x 4 k: the tail of every thread's 128 weights (NQ = 2)
    // The parts of a pair are blocks i + 8 p of a group of 8 NQ: blocks are dealt round-robin over the 8 XCDs, so all parts
    // share one XCD's L2 and the hand-off does not cross the fabric (speed only: nothing depends on it).
    const int grp = blockIdx.x / (8 * NQ), in_grp = blockIdx.x - grp * (8 * NQ);
    const int hf = in_grp >> 3, pair = grp * 8 + (in_grp & 7);  // pair = b * 2 + dir; hf = part
    if (pair >= n_pairs) return;                             // (padding blocks of the last group)
    if (drop_half && hf == NQ - 1) return;                   // (test hook: the partner that never shows up)
    const int b = pair >> 1, dir = pair & 1, tid = threadIdx.x;
    const int L = len_of(len, b);
    // Lane (row pair rp, K quarter kq): TWO rows x 64 k each.  Reading h from LDS is what a step costs most (round 4: with the
    // partner's values not even waited for, a step took 2.76 of its 2.94 us): every lane used to walk 128 k of ONE row, 32
    // 16-byte reads of h per step and lane; with two rows per lane a value of h read once feeds two rows: 16 reads.  The four
    // quarters of a row are the four lanes of a quad and meet in two DPP adds: (q0 + q1) + (q2 + q3).
    const int kq = tid & 3, rp = tid >> 2;                   // rows 2 rp, 2 rp + 1 of the part's ROWS (gate r / HU, unit r % HU)
    auto wrow = [&](int rr) { return (rr / HU) * 256 + hf * HU + (rr % HU); };  // row of W_hh / column of gx
    const int rowA = wrow(2 * rp), rowB = wrow(2 * rp + 1);
    // image [k4][row][4 k] (launch_transpose_whh): k4 = 16 kq + j covers the lane's 64 k
    const float4* W4 = reinterpret_cast<const float4*>(whhT + (long)dir * 256 * 1024) + (long)kq * 16 * 1024;
    float4 wreg[LSTMP_REG];  // pieces 2 j + s (s: row A / B), the first LSTMP_REG in registers, the rest in LDS
#pragma unroll
    for (int i = 0; i < LSTMP_REG; ++i) wreg[i] = W4[(long)(i >> 1) * 1024 + ((i & 1) ? rowB : rowA)];
#pragma unroll
    for (int i = 0; i < LSTMP_LDS; ++i) wl[i * NT + tid] = W4[(long)((LSTMP_REG + i) >> 1) * 1024 + ((i & 1) ? rowB : rowA)];
    if (tid < 4 * LSTMP_HP) hs[tid] = 0.f;
    if (tid == 0) *abort_flag = 0;
    float c = 0.f;
    // exchange slots: [pair][parity][256] granules, unit u of the hidden state at index u (part u / HU writes it)
    unsigned long long* slots = xchg + (long)pair * 2 * 256;  // + parity * 256
    // lanes kq = 0 / 1 of a quad finish row A / B: they carry its input projection
    const int my_rr = 2 * rp + (kq & 1);
    const float* gxp = gx + b * gx_bs + dir * 1024 + ((kq & 1) ? rowB : rowA);
    float gxv = (kq < 2 && L > 0) ? gxp[(long)(dir ? L - 1 : 0) * gx_ld] : 0.f;
    auto hslot = [](int k) { return (k >> 6) * LSTMP_HP + (k & 63); };  // where h[k] lives
    __syncthreads();
    for (int step = 0; step < L; ++step) {
        const int t = dir ? (L - 1 - step) : step;
        const float4* h4 = reinterpret_cast<const float4*>(hs + kq * LSTMP_HP);
        float accA = 0.f, accB = 0.f;
#pragma unroll
        for (int j = 0; j < LSTMP_REG / 2; ++j) {
            const float4 hv = h4[j];
            const float4 wa = wreg[2 * j], wb = wreg[2 * j + 1];
            accA = fmaf(wa.x, hv.x, accA);
            accB = fmaf(wb.x, hv.x, accB);
            accA = fmaf(wa.y, hv.y, accA);
            accB = fmaf(wb.y, hv.y, accB);
            accA = fmaf(wa.z, hv.z, accA);
            accB = fmaf(wb.z, hv.z, accB);
            accA = fmaf(wa.w, hv.w, accA);
            accB = fmaf(wb.w, hv.w, accB);
        }
#pragma unroll
        for (int j = 0; j < LSTMP_LDS / 2; ++j) {
            const float4 hv = h4[LSTMP_REG / 2 + j];
            const float4 wa = wl[(2 * j) * NT + tid], wb = wl[(2 * j + 1) * NT + tid];
            accA = fmaf(wa.x, hv.x, accA);
            accB = fmaf(wb.x, hv.x, accB);
            accA = fmaf(wa.y, hv.y, accA);
            accB = fmaf(wb.y, hv.y, accB);
            accA = fmaf(wa.z, hv.z, accA);
            accB = fmaf(wb.z, hv.z, accB);
            accA = fmaf(wa.w, hv.w, accA);
            accB = fmaf(wb.w, hv.w, accB);
        }
        // the four K quarters of a row: (q0 + q1) + (q2 + q3) in every lane of the quad
        accA += __shfl_xor(accA, 1);
        accB += __shfl_xor(accB, 1);
        accA += __shfl_xor(accA, 2);
        accB += __shfl_xor(accB, 2);
        if (kq < 2) {
            gates[my_rr] = gxv + ((kq & 1) ? accB : accA);
            // the input projection of the next step (its latency hides behind the rest of this one)
            if (step + 1 < L) gxv = gxp[(long)(dir ? t - 1 : t + 1) * gx_ld];
        }
        __syncthreads();
        const unsigned tag = (epoch << 16) | (unsigned)(step + 1);
        const int par = step & 1;
        if (tid < HU) {
            const float hn = lstm_cell(gates[tid], gates[HU + tid], gates[2 * HU + tid], gates[3 * HU + tid], c);
            if (step + 1 < L) {  // publish for the partners' next step first: one 8-byte agent-scope store
                const unsigned long long g8 = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(hn);
                __hip_atomic_store(slots + par * 256 + hf * HU + tid, g8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            hs[hslot(hf * HU + tid)] = hn;
            y[b * y_bs + (long)(dir * 256 + hf * HU + tid) * y_ld + t] = hn;
        } else if (tid < 256 && step + 1 < L) {
            // threads HU .. 255 fetch the partners' units: poll each granule until it carries this step's tag
            const int j = tid - HU;                        // 0 .. 255 - HU over the units that are not this part's
            const int u = j < hf * HU ? j : j + HU;
            unsigned long long g8 = 0;
            int spins = 0;
            for (;;) {
                g8 = __hip_atomic_load(slots + par * 256 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(g8 >> 32) == tag) break;
                if (++spins > spin_limit) {  // (1 << 19 ~ 0.2 s, far beyond any wait for a CU to free up: the partner is not coming)
                    atomicOr(err, 2u);
                    *abort_flag = 1;
                    g8 = 0;
                    break;
                }
                if (spins > 64) __builtin_amdgcn_s_sleep(1);  // (a tight poll while the partner is a fraction of a step away)
            }
            hs[hslot(u)] = __uint_as_float((unsigned)(g8 & 0xffffffffu));
        }
        __syncthreads();
        if (*abort_flag) return;  // (workgroup-uniform: written before the barrier above)
    }
}

// ---- the fall-back: ONE workgroup per (utterance, direction), most of W_hh streamed from L2 every step -------------------------
// No partner, no hand-off, nothing to wait for: a workgroup that cannot get a CU yet simply starts later.  It is what a model
// runs after a hand-off of the resident-weights forms timed out (a recurrence's partner starved behind other work on the GPU;
// DESIGN.md section 7), and since round 5 it gives the SAME BITS as they do: a lane is (row pair, K quarter) of the two-CU
// form and walks half 0's rows, then half 1's, with the same order of sums ((q0 + q1) + (q2 + q3) over the quad) and the same
// cell update -- so a model may switch between the forms at any call without its results changing (lstm_kernel above, whose
// lanes own whole rows, agrees with them to rounding only; it stays as KX_LSTM_PAIR=0's reference form).
// Half 0's first 8 pieces (of 16 per row) live in registers, its next 4 in LDS (128 KiB); the other 4 and all of half 1
// (62 % of the 1 MiB) come from L2 on every step: ~6 us per step against 1.6 - 2.5 us resident.
constexpr int LSTMS_REG = 16, LSTMS_LDS = 8;  // float4 pieces (2 j + row A / B) of half 0 in registers / in LDS
__global__ __launch_bounds__(1024) void lstm_stream_kernel(const float* gx, long gx_bs, int gx_ld, const float* whhT, float* y,
                                                           long y_bs, int y_ld, LenMap len) {
    extern __shared__ __attribute__((aligned(16))) float lstm_smem[];
    float* hs = lstm_smem;                                     // [4][LSTMP_HP]
    float* gates = hs + 4 * LSTMP_HP;                          // [1024] gate pre-activations, index = row of W_hh
    float4* wl = reinterpret_cast<float4*>(gates + 1024);      // [LSTMS_LDS][1024]
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int L = len_of(len, b);
    const int kq = tid & 3, rp = tid >> 2;
    // rows 2 rp, 2 rp + 1 of half p (the two-CU form's mapping: gate r >> 7, unit 128 p + (r & 127)); row B = row A + 1
    const int rr = 2 * rp;
    const int rowA0 = (rr >> 7) * 256 + (rr & 127), rowA1 = rowA0 + 128;
    const float4* W4 = reinterpret_cast<const float4*>(whhT + (long)dir * 256 * 1024) + (long)kq * 16 * 1024;
    float4 wreg[LSTMS_REG];
#pragma unroll
    for (int i = 0; i < LSTMS_REG; ++i) wreg[i] = W4[(long)(i >> 1) * 1024 + rowA0 + (i & 1)];
#pragma unroll
    for (int i = 0; i < LSTMS_LDS; ++i) wl[i * 1024 + tid] = W4[(long)((LSTMS_REG + i) >> 1) * 1024 + rowA0 + (i & 1)];
    if (tid < 4 * LSTMP_HP) hs[tid] = 0.f;
    float c = 0.f;
    // lanes kq = 0 / 1 of a quad finish row A / B of each half: they carry its input projection
    const int myrow0 = rowA0 + (kq & 1), myrow1 = rowA1 + (kq & 1);
    const float* gxp = gx + b * gx_bs + dir * 1024;
    float gxv0 = (kq < 2 && L > 0) ? gxp[(long)(dir ? L - 1 : 0) * gx_ld + myrow0] : 0.f;
    float gxv1 = (kq < 2 && L > 0) ? gxp[(long)(dir ? L - 1 : 0) * gx_ld + myrow1] : 0.f;
    auto hslot = [](int k) { return (k >> 6) * LSTMP_HP + (k & 63); };
    auto fma8 = [](const float4& wa, const float4& wb, const float4& hv, float& accA, float& accB) __attribute__((always_inline)) {
        accA = fmaf(wa.x, hv.x, accA);
        accB = fmaf(wb.x, hv.x, accB);
        accA = fmaf(wa.y, hv.y, accA);
        accB = fmaf(wb.y, hv.y, accB);
        accA = fmaf(wa.z, hv.z, accA);
        accB = fmaf(wb.z, hv.z, accB);
        accA = fmaf(wa.w, hv.w, accA);
        accB = fmaf(wb.w, hv.w, accB);
    };
    __syncthreads();
    for (int step = 0; step < L; ++step) {
        const int t = dir ? (L - 1 - step) : step;
        const float4* h4 = reinterpret_cast<const float4*>(hs + kq * LSTMP_HP);
        // half 1 first in program order (all of it streamed: its loads are in flight while half 0's resident part is summed)
        float a1 = 0.f, b1 = 0.f;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const float4 wa = W4[(long)j * 1024 + rowA1], wb = W4[(long)j * 1024 + rowA1 + 1];
            fma8(wa, wb, h4[j], a1, b1);
        }
        float a0 = 0.f, b0 = 0.f;
#pragma unroll
        for (int j = 0; j < LSTMS_REG / 2; ++j) fma8(wreg[2 * j], wreg[2 * j + 1], h4[j], a0, b0);
#pragma unroll
        for (int j = 0; j < LSTMS_LDS / 2; ++j) fma8(wl[(2 * j) * 1024 + tid], wl[(2 * j + 1) * 1024 + tid], h4[LSTMS_REG / 2 + j], a0, b0);
#pragma unroll
        for (int j = (LSTMS_REG + LSTMS_LDS) / 2; j < 16; ++j) {
            const float4 wa = W4[(long)j * 1024 + rowA0], wb = W4[(long)j * 1024 + rowA0 + 1];
            fma8(wa, wb, h4[j], a0, b0);
        }
        a0 += __shfl_xor(a0, 1);
        b0 += __shfl_xor(b0, 1);
        a0 += __shfl_xor(a0, 2);
        b0 += __shfl_xor(b0, 2);
        a1 += __shfl_xor(a1, 1);
        b1 += __shfl_xor(b1, 1);
        a1 += __shfl_xor(a1, 2);
        b1 += __shfl_xor(b1, 2);
        if (kq < 2) {
            gates[myrow0] = gxv0 + ((kq & 1) ? b0 : a0);
            gates[myrow1] = gxv1 + ((kq & 1) ? b1 : a1);
            if (step + 1 < L) {
                gxv0 = gxp[(long)(dir ? t - 1 : t + 1) * gx_ld + myrow0];
                gxv1 = gxp[(long)(dir ? t - 1 : t + 1) * gx_ld + myrow1];
            }
        }
        __syncthreads();
        if (tid < 256) {
            const float hn = lstm_cell(gates[tid], gates[256 + tid], gates[512 + tid], gates[768 + tid], c);
            hs[hslot(tid)] = hn;
            y[b * y_bs + (long)(dir * 256 + tid) * y_ld + t] = hn;
        }
        __syncthreads();
    }
}

// test hook (kx_test_lstm_fault): n > 0 = from the n-th launch of the two-CU kernel on, every launch loses the second half
// of each pair and polls with a short limit, so that the bounded wait's error path can be exercised (n = 6: the
// frame-axis LSTM of a forward, which runs after the forward's mid-way error check)
static std::atomic<int> lstm_test_fault{0};
void lstm_set_test_fault(int nth) { lstm_test_fault.store(nth > 0 ? nth : 0); }

// 0 = by batch size, 2 / 4 = that many workgroups per (utterance, direction), 1 = the streaming fall-back (KX_LSTM_PARTS; kx_test_lstm_parts)
static std::atomic<int> lstm_parts_force{getenv("KX_LSTM_PARTS") ? atoi(getenv("KX_LSTM_PARTS")) : 0};
void lstm_set_parts(int n) { lstm_parts_force.store(n == 1 || n == 2 || n == 4 ? n : 0); }  // (1 = the streaming fall-back)

static bool lstm_use_pair() {
    static const int v = getenv("KX_LSTM_PAIR") ? atoi(getenv("KX_LSTM_PAIR")) : 1;
    return v != 0;
}

size_t lstm_exchange_bytes(int B) { return (size_t)B * 2 * 2 * 2 * 128 * sizeof(unsigned long long); }

void launch_lstm(const float* gx, long gx_bs, int gx_ld, const float* whhT, float* y, long y_bs, int y_ld,
                 LenMap len, int B, unsigned long long* xchg, unsigned* err_word, hipStream_t s, unsigned* epoch_state) {
    static_assert(LSTM_LDS_K % 4 == 0 && LSTM_REG_K % 4 == 0, "whole float4 groups of h");
    if (lstm_use_pair() && xchg && err_word && lstm_parts_force.load() != 1) {
        // The tag's epoch is 16 bits wide and counted PER exchange buffer (epoch_state; the test hook's one-shot buffer has
        // none): when it wraps the buffer is cleared in stream order, so a granule of 65535 launches ago can never carry
        // the tag of a live step.  (0 is what a cleared buffer holds and is never used as an epoch.)
        static std::atomic<unsigned> hook_ctr{0};
        unsigned epoch;
        if (epoch_state) {
            epoch = (*epoch_state + 1) & 0xffffu;
            if (epoch == 0) {
                KX_HIP(hipMemsetAsync(xchg, 0, lstm_exchange_bytes(B), s));
                epoch = 1;
            }
            *epoch_state = epoch;
        } else {
            epoch = (hook_ctr.fetch_add(1) % 0xffffu) + 1;
        }
        const int n_pairs = B * 2;  // (utterance, direction); blocks come in groups of 16 = 8 pairs
        int fault = lstm_test_fault.load();
        if (fault > 1) {  // (count down to the launch that fails)
            lstm_test_fault.store(fault - 1);
            fault = 0;
        }
        // Four parts per (utterance, direction) while that leaves no CU without work: 8 B workgroups of 512 threads, each of which
        // needs a CU's whole register file (256 registers per lane).  Same bits either way (see lstm_pair_kernel).
        // KX_LSTM_PARTS = 2 / 4 forces a form.
        const int parts = lstm_parts_force.load();
        const bool four = parts ? parts == 4 : 8 * B <= conv16_cu_count();
        const int spin = fault ? (1 << 10) : (1 << 19);
        if (four) {
            using P4 = LstmParts<4>;
            const size_t lds = sizeof(float) * (4 * LSTMP_HP + P4::ROWS + 4 + (size_t)P4::LDSP * P4::THREADS * 4);
            hipLaunchKernelGGL(lstm_pair_kernel<4>, dim3(((n_pairs + 7) / 8) * 32), dim3(P4::THREADS), lds, s, gx, gx_bs, gx_ld, whhT, y,
                               y_bs, y_ld, len, xchg, epoch, err_word, n_pairs, spin, fault);
        } else {
            using P2 = LstmParts<2>;
            const size_t lds = sizeof(float) * (4 * LSTMP_HP + P2::ROWS + 4 + (size_t)P2::LDSP * P2::THREADS * 4);
            static DynLdsLimit pair_limit;  // (per device: each GPU's model launches from its own host thread)
            pair_limit.ensure(reinterpret_cast<const void*>(lstm_pair_kernel<2>), lds);
            hipLaunchKernelGGL(lstm_pair_kernel<2>, dim3(((n_pairs + 7) / 8) * 16), dim3(P2::THREADS), lds, s, gx, gx_bs, gx_ld, whhT, y,
                               y_bs, y_ld, len, xchg, epoch, err_word, n_pairs, spin, fault);
        }
        KX_HIP(hipGetLastError());
        return;
    }
    if (lstm_use_pair()) {  // the fall-back of a model whose hand-off timed out (xchg = null): same bits as the resident forms
        const size_t lds = sizeof(float) * (4 * LSTMP_HP + 1024 + (size_t)LSTMS_LDS * 1024 * 4);
        static DynLdsLimit stream_limit;
        stream_limit.ensure(reinterpret_cast<const void*>(lstm_stream_kernel), lds);
        hipLaunchKernelGGL(lstm_stream_kernel, dim3(B, 2), dim3(1024), lds, s, gx, gx_bs, gx_ld, whhT, y, y_bs, y_ld, len);
        KX_HIP(hipGetLastError());
        return;
    }
    const size_t lds = sizeof(float) * (256 + 1024 + (size_t)LSTM_LDS_K * 1024);
    static DynLdsLimit one_limit;
    one_limit.ensure(reinterpret_cast<const void*>(lstm_kernel), lds);
    hipLaunchKernelGGL(lstm_kernel, dim3(B, 2), dim3(1024), lds, s, gx, gx_bs, gx_ld, whhT, y, y_bs, y_ld, len);
    KX_HIP(hipGetLastError());
}

// ---- duration head + alignment ---------------------------------------------------------------
// dur[t] = clamp(round(sum_c sigmoid(logit[c][t]) / speed), min 1); frames = sum; idx = the
// repeat_interleave gather index (model.py forward_with_tokens).
__global__ __launch_bounds__(512) void duration_kernel(const float* logits, long bs, int ld, const float* speeds,
                                                       int n_speed, const int* lens, const int* pinned,
                                                       int n_pinned, int* dur, int* frames, int* idx, int idx_ld) {
    __shared__ int scan[512];
    const int b = blockIdx.x, t = threadIdx.x;
    const int T = lens[b];
    int d = 0;
    if (t < T) {
        float s = 0.f;
        for (int c = 0; c < 50; ++c) s += 1.0f / (1.0f + expf(-logits[b * bs + (long)c * ld + t]));
        s = s / speeds[n_speed > 1 ? b : 0];
        s = rintf(s);
        d = s < 1.f ? 1 : (int)s;
        if (n_pinned > 0) d = pinned[t % n_pinned];
        dur[b * 512 + t] = d;
    }
    scan[t] = d;
    __syncthreads();
    for (int o = 1; o < 512; o <<= 1) {
        const int v = (t >= o) ? scan[t - o] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    if (t == 511) frames[b] = scan[511];
    if (t < T) {
        const int start = scan[t] - d;
        for (int f = 0; f < d; ++f) idx[(long)b * idx_ld + start + f] = t;
    }
}
void launch_duration(const float* logits, long bs, int ld, const float* speeds, int n_speed, const int* lens,
                     const int* pinned, int n_pinned, int* dur, int* frames, int* idx, int idx_ld, int B,
                     hipStream_t s) {
    hipLaunchKernelGGL(duration_kernel, dim3(B), dim3(512), 0, s, logits, bs, ld, speeds, n_speed, lens, pinned,
                       n_pinned, dur, frames, idx, idx_ld);
    KX_HIP(hipGetLastError());
}

__global__ void gather_cols_kernel(const float* src, long sbs, int sld, float* dst, long dbs, int dld,
                                   const int* idx, int idx_ld, const int* frames) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (f >= frames[b]) return;
    dst[b * dbs + (long)c * dld + f] = src[b * sbs + (long)c * sld + idx[(long)b * idx_ld + f]];
}
void launch_gather_cols(const float* src, long sbs, int sld, float* dst, long dbs, int dld, int C, const int* idx,
                        int idx_ld, const int* frames, int B, int Fmax, hipStream_t s) {
    hipLaunchKernelGGL(gather_cols_kernel, dim3((Fmax + 255) / 256, C, B), dim3(256), 0, s, src, sbs, sld, dst, dbs,
                       dld, idx, idx_ld, frames);
    KX_HIP(hipGetLastError());
}

// ---- AdainResBlk1d up-sampling "pool" ---------------------------------------------------------
// depth-wise ConvTranspose1d(k3, s2, p1, output_padding 1) over v = leaky(adain(x)):
//   out[2m] = v[m] w1 + b ;  out[2m+1] = v[m] w2 + v[m+1] w0 + b
__global__ void pool_up2_kernel(const float* x, long xbs, int xld, const float* mean, const float* scale,
                                const float* shift, int n_bs, float slope, const float* w, const float* bias,
                                float* y, long ybs, int yld, LenMap in_len) {
    const int t2 = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int L = len_of(in_len, b);
    if (t2 >= 2 * L) return;
    const float mu = mean[(long)b * n_bs + c], sc = scale[(long)b * n_bs + c], sh = shift[(long)b * n_bs + c];
    const float* row = x + b * xbs + (long)c * xld;
    const int m = t2 >> 1;
    float v0 = (row[m] - mu) * sc + sh;
    v0 = v0 > 0.f ? v0 : v0 * slope;
    float o;
    if ((t2 & 1) == 0) {
        o = v0 * w[c * 3 + 1];
    } else {
        float v1 = 0.f;
        if (m + 1 < L) {
            v1 = (row[m + 1] - mu) * sc + sh;
            v1 = v1 > 0.f ? v1 : v1 * slope;
        }
        o = v0 * w[c * 3 + 2] + v1 * w[c * 3 + 0];
    }
    y[b * ybs + (long)c * yld + t2] = o + bias[c];
}
void launch_pool_up2(const float* x, long xbs, int xld, int C, const float* mean, const float* scale,
                     const float* shift, int n_bs, float slope, const float* w, const float* bias, float* y,
                     long ybs, int yld, LenMap in_len, int B, int Lmax_in, hipStream_t s) {
    hipLaunchKernelGGL(pool_up2_kernel, dim3((2 * Lmax_in + 255) / 256, C, B), dim3(256), 0, s, x, xbs, xld, mean,
                       scale, shift, n_bs, slope, w, bias, y, ybs, yld, in_len);
    KX_HIP(hipGetLastError());
}

// ---- harmonic source (SourceModuleHnNSF / SineGen, istftnet.py) ---------------------------
// The phase path is bit-exact with torch CPU float32: rad = (f0*h/24000) mod 1, cumulative
// sum accumulated in float64 and rounded per step (torch.cumsum on CPU), x 2 x pi x 300, then
// the x300 linear up-sampling  src = fma(1/300, j+0.5, -0.5);  out = fma(l0, p0, l1*p1).
// Phases reach ~1e5 rad, where one float32 ulp is ~0.008 rad, so any other operation order
// decorrelates the harmonics ("Parity").
// Three passes per block of SRC_CH steps (round 5): the per-step terms rad for all nine harmonics in parallel, then the running sums
// by nine lanes -- the ONLY sequential part: one f64 add per step, in step order, exactly the chain of the one-loop form it
// replaces (201 us for 844 steps at batch 1: fmodf, a division and a global load inside the dependent loop) -- then the scaling
// in parallel.  Same operations on the same values in the same order: same bits.
constexpr int SRC_CH = 1024;
__global__ __launch_bounds__(256) void source_phase_kernel(const float* f0, long f0_bs, const int* frames, float* phase, int L2max) {
    __shared__ float buf[9][SRC_CH + 1];  // (+ 1: the nine lanes of the second pass walk nine rows side by side)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n2 = 2 * frames[b];
    double cs = 0.0;  // (lanes 0 .. 8: the running sum of harmonic tid)
    for (int base = 0; base < n2; base += SRC_CH) {
        const int n = n2 - base < SRC_CH ? n2 - base : SRC_CH;
        for (int i = tid; i < n; i += 256) {
            const float f = f0[b * f0_bs + base + i];
#pragma unroll
            for (int h = 0; h < 9; ++h) {
                const float fn = __fmul_rn(f, (float)(h + 1));
                float rad = fmodf(__fdiv_rn(fn, 24000.0f), 1.0f);
                if (rad < 0.f) rad = __fadd_rn(rad, 1.0f);
                buf[h][i] = rad;
            }
        }
        __syncthreads();
        if (tid < 9) {
            for (int i = 0; i < n; ++i) {
                cs += (double)buf[tid][i];
                buf[tid][i] = (float)cs;
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
#pragma unroll
            for (int h = 0; h < 9; ++h)
                phase[((long)b * 9 + h) * L2max + base + i] = __fmul_rn(__fmul_rn(__fmul_rn(buf[h][i], 2.0f), 3.14159274101257324f), 300.0f);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t& o0, uint32_t& o1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0;
    o1 = c1;
}

__global__ void source_sample_kernel(const float* f0, long f0_bs, const int* frames, const float* phase, int L2max,
                                     const float* lin_w, const float* lin_b, uint32_t k0, uint32_t k1,
                                     uint64_t utt_base, const uint64_t* utt_seeds, int noise_off, float* har,
                                     long har_bs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const int n2 = 2 * frames[b];
    if (j >= 300 * n2) return;
    const float f0v = f0[b * f0_bs + j / 300];
    const bool uv = f0v > 10.0f;
    const float scale = (float)(1.0 / 300.0);
    float src = fmaf(scale, (float)j + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src;
    const float l1 = __fsub_rn(src, (float)i0), l0 = __fsub_rn(1.0f, l1);
    const int i1 = i0 + (i0 < n2 - 1 ? 1 : 0);
    const float amp = uv ? 0.003f : (0.1f / 3.0f);
    // per-utterance keys (dispatcher): stream (seed_b, utterance 0); otherwise (seed, utt_base + b)
    uint32_t utt = (uint32_t)(utt_base + (uint64_t)b);
    if (utt_seeds) {
        k0 = (uint32_t)(utt_seeds[b] & 0xFFFFFFFFu);
        k1 = (uint32_t)(utt_seeds[b] >> 32);
        utt = 0u;
    }
    float acc = 0.f;
#pragma unroll
    for (int h = 0; h < 9; ++h) {
        const float* P = phase + ((long)b * 9 + h) * L2max;
        const float ph = fmaf(l0, P[i0], __fmul_rn(l1, P[i1]));
        const float sv = sinf(ph) * 0.1f;
        float nz = 0.f;
        if (!noise_off) {
            uint32_t x0, x1;
            philox4x32_10((uint32_t)j, (uint32_t)h, utt, 0u, k0, k1, x0, x1);
            const float u1 = ((float)(x0 >> 9) + 0.5f) * 1.1920928955078125e-07f;
            const float u2 = ((float)(x1 >> 9) + 0.5f) * 1.1920928955078125e-07f;
            const float rr = sqrtf(-2.0f * logf(u1));
            nz = amp * (rr * cosf(6.2831855f * u2));
        }
        const float sw = uv ? (sv + nz) : nz;
        acc += lin_w[h] * sw;
    }
    har[b * har_bs + j] = tanhf(acc + lin_b[0]);
}

void launch_source(const float* f0, long f0_bs, const int* frames, int B, int Fmax, const float* lin_w,
                   const float* lin_b, uint64_t seed, uint64_t utt_base, const uint64_t* utt_seeds, int noise_off,
                   float* phase_ws, float* har, long har_bs, hipStream_t s) {
    const int L2max = 2 * Fmax;
    hipLaunchKernelGGL(source_phase_kernel, dim3(B), dim3(256), 0, s, f0, f0_bs, frames, phase_ws, L2max);
    KX_HIP(hipGetLastError());
    hipLaunchKernelGGL(source_sample_kernel, dim3((600 * Fmax + 255) / 256, B), dim3(256), 0, s, f0, f0_bs, frames,
                       phase_ws, L2max, lin_w, lin_b, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32),
                       utt_base, utt_seeds, noise_off, har, har_bs);
    KX_HIP(hipGetLastError());
}

// ---- STFT (n_fft 20, hop 5, periodic Hann, replicate centre padding) and iSTFT head -------
// Two variants of the pair, selected per model (kx_set_stft_variant):
//   0 "onnx"  : the conv-based pair the ONNX export uses (upstream custom_stft.py, as recalled - SURVEY Appendix A):
//               magnitude sqrt(re^2 + im^2 + 1e-14); phase atan2 with (im == 0 && re < 0) forced to +pi; inverse =
//               two transposed convolutions with cos/sin * window / n_fft summed as real - imag: NO doubling of the
//               interior one-sided bins and NO division by the window envelope.  This is the graph `sess.run`
//               executes (kokorox/src/onn/ort_koko.rs:79), hence the default.
//   1 "torch" : torch.stft / torch.istft semantics (one-sided irfft with doubled interior bins, overlap-add
//               divided by the summed squared window) - what the PyTorch model does with disable_complex=False.
__constant__ float c_fwd_re[11][20];
__constant__ float c_fwd_im[11][20];
__constant__ float c_inv_re[2][20][11];
__constant__ float c_inv_im[2][20][11];
__constant__ float c_win_sq[20];

void init_dft_tables() {
    double c[20], sn[20], win[20];
    for (int m = 0; m < 20; ++m) {
        c[m] = cos(2.0 * M_PI * m / 20.0);
        sn[m] = sin(2.0 * M_PI * m / 20.0);
        win[m] = 0.5 - 0.5 * cos(2.0 * M_PI * m / 20.0);
    }
    c[0] = 1; c[5] = 0; c[10] = -1; c[15] = 0;
    sn[0] = 0; sn[5] = 1; sn[10] = 0; sn[15] = -1;
    float fr[11][20], fi[11][20], ir[2][20][11], ii[2][20][11], wsq[20];
    for (int k = 0; k < 11; ++k)
        for (int n = 0; n < 20; ++n) {
            const int m = (k * n) % 20;
            fr[k][n] = (float)(win[n] * c[m]);
            fi[k][n] = (float)(win[n] * (-sn[m]));
            // variant 0: cos / sin * window / n_fft for every bin, combined as real - imag
            ir[0][n][k] = (float)(c[m] / 20.0 * win[n]);
            ii[0][n][k] = (float)(-sn[m] / 20.0 * win[n]);
            // variant 1: one-sided irfft (interior bins doubled, imaginary parts of bins 0 and 10 ignored)
            const double ck = (k == 0 || k == 10) ? 1.0 : 2.0;
            ir[1][n][k] = (float)((ck * c[m]) / 20.0 * win[n]);
            ii[1][n][k] = (k == 0 || k == 10) ? 0.f : (float)((-ck * sn[m]) / 20.0 * win[n]);
        }
    for (int n = 0; n < 20; ++n) wsq[n] = (float)(win[n] * win[n]);
    KX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_fwd_re), fr, sizeof(fr)));
    KX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_fwd_im), fi, sizeof(fi)));
    KX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_inv_re), ir, sizeof(ir)));
    KX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_inv_im), ii, sizeof(ii)));
    KX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_win_sq), wsq, sizeof(wsq)));
}

__global__ void stft_kernel(const float* hs, long hs_bs, float* har, long bs, int ld, const int* frames, int variant) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const int L = 600 * frames[b], nf = 120 * frames[b] + 1;
    if (f >= nf) return;
    float x[20];
#pragma unroll
    for (int n = 0; n < 20; ++n) {
        int p = 5 * f - 10 + n;
        p = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);
        x[n] = hs[b * hs_bs + p];
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int n = 0; n < 20; ++n) {
            re += x[n] * c_fwd_re[k][n];
            im += x[n] * c_fwd_im[k][n];
        }
        float mag, ph;
        if (variant == 0) {
            mag = sqrtf(re * re + im * im + 1e-14f);
            ph = (im == 0.f && re < 0.f) ? 3.14159274101257324f : atan2f(im, re);
        } else {
            mag = sqrtf(re * re + im * im);
            ph = atan2f(im, re);
        }
        har[b * bs + (long)k * ld + f] = mag;
        har[b * bs + (long)(11 + k) * ld + f] = ph;
    }
}
void launch_stft(const float* har_src, long hs_bs, float* har, long bs, int ld, const int* frames, int B, int Fmax,
                 int variant, hipStream_t s) {
    hipLaunchKernelGGL(stft_kernel, dim3((120 * Fmax + 1 + 255) / 256, B), dim3(256), 0, s, har_src, hs_bs, har, bs,
                       ld, frames, variant);
    KX_HIP(hipGetLastError());
}

// head: mag = exp(x[:11]), phase = sin(x[11:]) (Generator.forward), then the inverse of the chosen variant
__global__ void istft_spec_kernel(const float* cp, long bs, int ld, float* spec, const int* frames) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y, b = blockIdx.z;
    const int nf = 120 * frames[b] + 1;
    if (f >= nf) return;
    const float mag = expf(cp[b * bs + (long)k * ld + f]);
    const float ph = sinf(cp[b * bs + (long)(11 + k) * ld + f]);
    spec[b * bs + (long)k * ld + f] = mag * cosf(ph);
    spec[b * bs + (long)(11 + k) * ld + f] = mag * sinf(ph);
}
__global__ void istft_ola_kernel(const float* spec, long bs, int ld, float* audio, long audio_ld, const int* frames,
                                 int variant) {
    // The inverse-DFT rows are indexed by the sample's position in its frame, which differs from lane to lane: from constant
    // memory that is one scalar load per distinct index (808 us per batch-64 launch); from LDS it is an ordinary banked read.
    // The spectrum columns a block's samples touch are staged in LDS too: every sample reads 88 of them, five neighbouring
    // samples the same ones (407 -> 248 us).  And a thread takes one HOP of five consecutive samples (n = 5 q + 10 .. + 4: the
    // same four frames q - 1 .. q + 2 for all five), so the 88 spectrum values are read once per five samples and every
    // coefficient read is the same address in all lanes (a broadcast): 248 -> 196 us.  Per sample the frames are still added
    // in descending order and the bins in ascending order, as before.
    constexpr int HPB = 256;      // hops (threads) per block = 1280 samples
    constexpr int FW = HPB + 4;   // frames staged: hop q touches frames q - 1 .. q + 2
    __shared__ float s_re[20][11], s_im[20][11], s_wsq[20];
    __shared__ float s_sp[22][FW];
    const int b = blockIdx.y, q0 = blockIdx.x * HPB;
    const int nf = 120 * frames[b] + 1, ns = 600 * frames[b];
    if (5 * q0 >= ns) return;  // (block-uniform)
    for (int i = threadIdx.x; i < 220; i += blockDim.x) {
        s_re[i / 11][i % 11] = c_inv_re[variant][i / 11][i % 11];
        s_im[i / 11][i % 11] = c_inv_im[variant][i / 11][i % 11];
    }
    if (threadIdx.x < 20) s_wsq[threadIdx.x] = c_win_sq[threadIdx.x];
    const int f0 = q0 - 1;  // (may be -1: staged as zero, never read)
    const float* sp = spec + b * bs;
    for (int i = threadIdx.x; i < 22 * FW; i += blockDim.x) {
        const int k = i / FW, f = f0 + i % FW;
        s_sp[k][i % FW] = (f >= 0 && f < nf) ? sp[(long)k * ld + f] : 0.f;
    }
    __syncthreads();
    const int q = q0 + threadIdx.x;  // samples j = 5 q .. 5 q + 4, n = j + 10 = 5 (q + 2) + s
    if (5 * q >= ns) return;
    float y[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, env[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    // sample s of the hop: fhi = (n / 5) = q + 2, flo = (n - 15) / 5 = q - 1 (n >= 19) -- clamped as before
#pragma unroll
    for (int df = 3; df >= 0; --df) {  // frame f = q - 1 + df, descending
        const int f = q - 1 + df;
        if (f > nf - 1 || f < 0) continue;
        float re[11], im[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            re[k] = s_sp[k][f - f0];
            im[k] = s_sp[11 + k][f - f0];
        }
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            const int n = 5 * q + 10 + s5;
            const int m = n - 5 * f;  // = 5 (3 - df) + s5: a compile-time constant once unrolled
            // (the first samples of the signal, n < 19, reach back only to frame 0: f >= 0 above enforces it)
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) acc += s_re[m][k] * re[k] + s_im[m][k] * im[k];
            y[s5] += acc;
            env[s5] += s_wsq[m];
        }
    }
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
        const int j = 5 * q + s5;
        if (j < ns) audio[b * audio_ld + j] = variant == 0 ? y[s5] : y[s5] / env[s5];
    }
}
void launch_istft_head(const float* cp, long bs, int ld, float* spec_ws, float* audio, long audio_ld,
                       const int* frames, int B, int Fmax, int variant, hipStream_t s) {
    const int nfmax = 120 * Fmax + 1;
    hipLaunchKernelGGL(istft_spec_kernel, dim3((nfmax + 255) / 256, 11, B), dim3(256), 0, s, cp, bs, ld, spec_ws,
                       frames);
    KX_HIP(hipGetLastError());
    hipLaunchKernelGGL(istft_ola_kernel, dim3((120 * Fmax + 255) / 256, B), dim3(256), 0, s, spec_ws, bs, ld, audio,
                       audio_ld, frames, variant);  // (one thread per hop of five samples)
    KX_HIP(hipGetLastError());
}

}  // namespace kx
