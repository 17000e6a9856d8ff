// Probe: do vector instructions hide behind v_mfma_f32_32x32x16_f16 on gfx950 with TWO waves per SIMD (the direct-A conv's
// occupancy), and does it matter whether the accumulators live in ArchVGPRs or AccVGPRs?
// Each wave: 8 accumulator tiles, 3 dependent MFMAs per tile and step (as the f16x3 conv), NV independent v_fma_f32 per MFMA
// (own registers, no memory).  Reports shader cycles per MFMA and wave.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;


// One half-unit pair of the conv's input transform (InstanceNorm affine + snake + f16 hi/lo split + two LDS dwords), as the
// compiler emits it, one instruction per call: op<I>() for I = 0 .. 53.  State in xs[].
struct XState { float x, t, n, z, p, y0, y1; unsigned hp, lp; float prm; unsigned addr; };
template <int I>
__device__ __forceinline__ void xop(XState& s) {
    constexpr int J = I % 27;       // element 0: J of 0..26 on y0; element 1 on y1, then the pack
    constexpr bool second = I >= 27;
    float& y = second ? s.y1 : s.y0;
    int sg;
    if constexpr (J == 0) asm volatile("v_readlane_b32 %0, %1, 3\n\tv_subrev_f32 %2, %0, %2" : "=&s"(sg), "+v"(s.prm), "+v"(s.x));
    else if constexpr (J == 1) asm volatile("v_readlane_b32 %0, %1, 4\n\tv_mul_f32 %2, %0, %2" : "=&s"(sg), "+v"(s.prm), "+v"(s.x));
    else if constexpr (J == 2) asm volatile("v_readlane_b32 %0, %1, 5\n\tv_mul_f32 %2, %0, %3" : "=&s"(sg), "+v"(s.prm), "=v"(s.t) : "v"(s.x));
    else if constexpr (J == 3) asm volatile("v_mul_f32 %0, 0x3ea2f983, %1" : "=v"(s.n) : "v"(s.t));
    else if constexpr (J == 4) asm volatile("v_rndne_f32 %0, %0" : "+v"(s.n));
    else if constexpr (J == 5) asm volatile("v_fmac_f32 %0, 0xc0490fdb, %1" : "+v"(s.t) : "v"(s.n));
    else if constexpr (J == 6) asm volatile("v_fmac_f32 %0, 0x33bbbd2e, %1" : "+v"(s.t) : "v"(s.n));
    else if constexpr (J == 7) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(s.z) : "v"(s.t));
    else if constexpr (J == 8) asm volatile("v_fmamk_f32 %0, %1, 0xb672eaaa, %2" : "=v"(s.p) : "v"(s.z), "v"(s.prm));
    else if constexpr (J == 9) asm volatile("v_fmaak_f32 %0, %1, %0, 0xbb4fe5f6" : "+v"(s.p) : "v"(s.z));
    else if constexpr (J == 10) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3d3609f3" : "+v"(s.p) : "v"(s.z));
    else if constexpr (J == 11) asm volatile("v_fmaak_f32 %0, %1, %0, 0xbeaaaaa1" : "+v"(s.p) : "v"(s.z));
    else if constexpr (J == 12) asm volatile("v_fma_f32 %0, %1, %0, 1.0" : "+v"(s.p) : "v"(s.z));
    else if constexpr (J == 13) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(s.p) : "v"(s.z));
    else if constexpr (J == 14) asm volatile("v_readlane_b32 %0, %1, 11\n\tv_fmac_f32 %2, %0, %3" : "=&s"(sg), "+v"(s.prm), "+v"(s.x) : "v"(s.p));
    else if constexpr (J == 15) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y) : "v"(s.prm), "v"(s.x));
    else if constexpr (J == 16) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(y) : "v"(s.prm), "v"(s.t));
    else if constexpr (J == 17 && second) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(s.hp) : "v"(s.y0), "v"(s.y1));
    else if constexpr (J == 18 && second) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(s.t) : "v"(s.hp));
    else if constexpr (J == 19 && second) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.y0) : "v"(s.t));
    else if constexpr (J == 20 && second) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(s.t) : "v"(s.hp));
    else if constexpr (J == 21 && second) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.y1) : "v"(s.t));
    else if constexpr (J == 22 && second) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(s.lp) : "v"(s.y0), "v"(s.y1));
    else if constexpr (J == 23 && second) asm volatile("ds_write_b32 %0, %1" ::"v"(s.addr), "v"(s.hp) : "memory");
    else if constexpr (J == 24 && second) asm volatile("ds_write_b32 %0, %1 offset:4096" ::"v"(s.addr), "v"(s.lp) : "memory");
    else if constexpr (J == 25 && second) asm volatile("v_mov_b32 %0, %1" : "=v"(s.x) : "v"(s.y1));
}
template <int LO, int HI, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); sfor<LO + 1, HI>(f); }
}

template <int NV, bool AGPR, int KIND>
__global__ __launch_bounds__(256, 2) void probe(const uint4* src, int steps, float* sink, unsigned long long* cyc) {
    __shared__ unsigned lds_buf[256 * 8];
    const int tid = threadIdx.x;
    half8 a = __builtin_bit_cast(half8, src[tid]), b = __builtin_bit_cast(half8, src[tid + 256]);
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)(tid + i) * 1e-3f;
    const float m = 1.0001f, c = 1e-4f;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 pv[4], pm = {m, m}, pc = {c, c};
#pragma unroll
    for (int i = 0; i < 4; ++i) pv[i] = f32x2{v[2 * i], v[2 * i + 1]};
    const unsigned long long c0 = __builtin_readcyclecounter();
    using f32x4p = __attribute__((ext_vector_type(4))) float;
    f32x4p acc4[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc4[j] = f32x4p{0.f, 0.f, 0.f, 0.f};
    XState xs{v[0], v[1], v[2], v[3], v[4], v[5], v[6], 0u, 0u, v[7], (unsigned)(tid * 4)};
    for (int s = 0; s < steps; ++s) {
        if constexpr (KIND == 11) {  // the same transform stream beside the same FLOPs issued as v_mfma_f32_16x16x32_f16 (two per slot)
            sfor<0, 24>([&](auto ic) {
                constexpr int idx = decltype(ic)::value;
                constexpr int blk = 4 * (idx / 3) + 2 * (idx & 1);
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[blk]) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[blk + 1]) : "v"(a), "v"(b));
                sfor<0, NV>([&](auto qc) { xop<(idx * NV + decltype(qc)::value) % 54>(xs); });
            });
            continue;
        }
        if constexpr (KIND == 10) {  // the conv's transform, NV of its instructions behind every MFMA, in program order
            sfor<0, 24>([&](auto ic) {
                constexpr int idx = decltype(ic)::value;
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[idx / 3]) : "v"(a), "v"(b));
                sfor<0, NV>([&](auto qc) { xop<(idx * NV + decltype(qc)::value) % 54>(xs); });
            });
            continue;
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(k * NV + q) & 7]) : "v"(m), "v"(c));
                    else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(m), "v"(c));  // one dependent chain
                    else if constexpr (KIND == 2) {  // v_readlane -> SGPR -> vector use (the transform's per-channel parameters)
                        int sg;
                        asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(v[(q + 1) & 7]));
                        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[q & 7]) : "s"(sg));
                        ++q;
                    } else if constexpr (KIND == 3) {
                        unsigned pk;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[q & 7]), "v"(v[(q + 1) & 7]));
                        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[q & 7]) : "v"(pk));
                        ++q;
                    } else if constexpr (KIND == 4) {
                        asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(tid * 16 + (q & 3) * 4)), "v"(v[q & 7]) : "memory");
                    } else if constexpr (KIND == 5) {  // independent packed f32 fma (two elements per instruction)
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[(k * NV + q) & 3]) : "v"(pm), "v"(pc));
                    } else if constexpr (KIND == 6) {  // one dependent packed chain
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[0]) : "v"(pm), "v"(pc));
                    } else if constexpr (KIND == 7) {  // two dependent chains, interleaved
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 1]) : "v"(m), "v"(c));
                    } else if constexpr (KIND == 8) {  // four dependent chains, interleaved
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 3]) : "v"(m), "v"(c));
                    } else if constexpr (KIND == 9) {  // two dependent packed chains, interleaved
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[q & 1]) : "v"(pm), "v"(pc));
                    }
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[j][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) t += v[i];
    t += xs.x + xs.y0 + xs.y1 + (float)xs.hp + (float)xs.lp;
    if constexpr (KIND == 11) {
#pragma unroll
        for (int j = 0; j < 32; ++j) t += acc4[j][0] + acc4[j][1] + acc4[j][2] + acc4[j][3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) t += pv[i][0] + pv[i][1];
    if (t == 123.456f) sink[0] = t;
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int NV, bool AGPR, int KIND>
static void run(const uint4* src, int steps, float* sink, unsigned long long* cyc, int n_wg) {
    hipLaunchKernelGGL((probe<NV, AGPR, KIND>), dim3(n_wg), dim3(256), 0, 0, src, steps, sink, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NV, AGPR, KIND>), dim3(n_wg), dim3(256), 0, 0, src, steps, sink, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = (double)n_wg * 4 * steps * 24 * 32768.0 / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h(n_wg);
    hipMemcpy(h.data(), cyc, n_wg * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : h) s += (double)x;
    const double per = s / n_wg / ((double)steps * 24);
    const char* kinds[12] = {"independent v_fma", "ONE dependent v_fma chain", "v_readlane + v_mul with that SGPR (pairs)", "v_cvt_pk_f16_f32 + v_cvt_f32_f16 (pairs)", "ds_write_b32",
                             "independent v_pk_fma_f32", "ONE dependent v_pk_fma_f32 chain", "TWO interleaved dependent v_fma chains", "FOUR interleaved dependent v_fma chains", "TWO interleaved dependent v_pk_fma_f32 chains", "the conv transform's own instruction sequence", "that sequence beside 2 x v_mfma_f32_16x16x32_f16 per slot"};
    printf("%s accumulators, %d x [%s] per MFMA: %.1f ticks per MFMA and wave; %d workgroups %.3f ms %.0f TFLOP/s issued\n", AGPR ? "AccVGPR " : "ArchVGPR", NV, kinds[KIND], per, n_wg, ms, tflops);
}

int main() {
    const int n_wg = 512, steps = 400;
    uint4* src;
    float* sink;
    unsigned long long* cyc;
    std::vector<unsigned short> h(512 * 8);
    srand(1);
    for (auto& x : h) x = (unsigned short)(((rand() & 1) << 15) | ((12 + rand() % 5) << 10) | (rand() & 1023));
    hipMalloc(&src, h.size() * 2);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, n_wg * 8);
    run<0, false, 0>(src, steps, sink, cyc, n_wg);
    run<4, false, 0>(src, steps, sink, cyc, n_wg);
    run<8, false, 0>(src, steps, sink, cyc, n_wg);
    run<4, true, 0>(src, steps, sink, cyc, n_wg);
    run<2, false, 1>(src, steps, sink, cyc, n_wg);
    run<4, false, 1>(src, steps, sink, cyc, n_wg);
    run<2, false, 2>(src, steps, sink, cyc, n_wg);
    run<4, false, 2>(src, steps, sink, cyc, n_wg);
    run<2, false, 3>(src, steps, sink, cyc, n_wg);
    run<4, false, 3>(src, steps, sink, cyc, n_wg);
    run<1, false, 4>(src, steps, sink, cyc, n_wg);
    run<2, false, 4>(src, steps, sink, cyc, n_wg);
    run<2, false, 5>(src, steps, sink, cyc, n_wg);
    run<4, false, 5>(src, steps, sink, cyc, n_wg);
    run<2, false, 6>(src, steps, sink, cyc, n_wg);
    run<4, false, 6>(src, steps, sink, cyc, n_wg);
    run<4, false, 7>(src, steps, sink, cyc, n_wg);
    run<6, false, 7>(src, steps, sink, cyc, n_wg);
    run<4, false, 8>(src, steps, sink, cyc, n_wg);
    run<6, false, 8>(src, steps, sink, cyc, n_wg);
    run<8, false, 8>(src, steps, sink, cyc, n_wg);
    run<4, false, 9>(src, steps, sink, cyc, n_wg);
    run<3, false, 1>(src, steps, sink, cyc, n_wg);
    run<6, false, 0>(src, steps, sink, cyc, n_wg);
    run<0, false, 0>(src, steps, sink, cyc, n_wg);
    run<0, false, 0>(src, steps, sink, cyc, 256);
    run<4, false, 0>(src, steps, sink, cyc, 256);
    run<0, false, 0>(src, steps * 8, sink, cyc, 512);
    run<4, false, 0>(src, steps * 8, sink, cyc, 512);
    run<0, false, 0>(src, steps * 8, sink, cyc, 2048);
    run<4, false, 0>(src, steps * 8, sink, cyc, 2048);
    run<2, false, 10>(src, steps, sink, cyc, n_wg);
    run<3, false, 10>(src, steps, sink, cyc, n_wg);
    run<4, false, 10>(src, steps, sink, cyc, n_wg);
    run<6, false, 10>(src, steps, sink, cyc, n_wg);
    run<9, false, 10>(src, steps, sink, cyc, n_wg);
    // long launches (the short ones carry a few % of noise): 32x32x16 against 16x16x32 with the transform stream of k = 11 / 7 / 3
    run<0, false, 10>(src, steps * 8, sink, cyc, n_wg);
    run<0, false, 11>(src, steps * 8, sink, cyc, n_wg);
    run<3, false, 10>(src, steps * 8, sink, cyc, n_wg);
    run<3, false, 11>(src, steps * 8, sink, cyc, n_wg);
    run<4, false, 10>(src, steps * 8, sink, cyc, n_wg);
    run<4, false, 11>(src, steps * 8, sink, cyc, n_wg);
    run<9, false, 10>(src, steps * 8, sink, cyc, n_wg);
    run<9, false, 11>(src, steps * 8, sink, cyc, n_wg);
    return 0;
}
