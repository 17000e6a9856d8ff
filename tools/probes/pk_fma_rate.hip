// Probe: is v_pk_fma_f32 full rate on gfx950 when nothing else competes (the LSTM recurrence's dot products)?  16 waves per CU
// (the LSTM kernel's 1024 threads), each a chain of independent accumulators; N scalar v_fma_f32 against N / 2 v_pk_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int PK>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    float a[8];
    f32x2 p[4];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    const float m = 1.0001f, c = 1e-4f;
    const f32x2 pm = {m, m}, pc = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (PK) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += a[i];
    for (int i = 0; i < 4; ++i) t += p[i][0] + p[i][1];
    out[blockIdx.x * 1024 + threadIdx.x] = t;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    for (int pk = 0; pk < 2; ++pk) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int iters = 20000;
        auto launch = [&]() {
            if (pk) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(1024), 0, 0, out, iters);
            else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(1024), 0, 0, out, iters);
        };
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fma = 256.0 * 1024 * iters * 16 * 8;
        printf("%s: %.3f ms, %.1f TFLOP/s f32 (2 per FMA)\n", pk ? "v_pk_fma_f32 (4 x 2)" : "v_fma_f32 (8)", ms, 2 * fma / (ms * 1e-3) / 1e12);
    }
    return 0;
}
