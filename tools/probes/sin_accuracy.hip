// Probe: absolute error of sin^2 evaluated three ways on gfx950, against f64.
//   (a) period-pi Cody-Waite + degree-5 polynomial (what conv_f16x3.hip uses), (b) 0.5 - 0.5 * v_cos_f32(t / pi turns), (c) v_sin_f32 squared
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ float sin_sq_poly(float t) {  // the form used by conv_f16x3.hip
    const float n = rintf(t * 0.318309886183790672f);
    float r = fmaf(n, -3.14159274101257324f, t);
    r = fmaf(n, 8.74227765734758578e-08f, r);
    const float z = r * r;
    float p = fmaf(z, -3.6197402550897095e-06f, 1.3928599946666651e-04f);
    p = fmaf(z, p, -3.1722760759294033e-03f);
    p = fmaf(z, p, 4.4443082064390182e-02f);
    p = fmaf(z, p, -3.3333304524421692e-01f);
    p = fmaf(z, p, 1.0f);
    return z * p;
}
__global__ void k(const float* t, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = t[i];
    o[i] = sin_sq_poly(x);
    o[n + i] = fmaf(-0.5f, __builtin_amdgcn_cosf(x * 0.318309886183790672f), 0.5f);  // cos(2x) = cos(2 pi * x/pi)
    const float sv = __builtin_amdgcn_sinf(x * 0.159154943091895336f);
    o[2 * n + i] = sv * sv;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = -40.f + 80.f * (float)i / n;  // |alpha * y| range of the snake arguments
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 3 * n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, o, n);
    std::vector<float> r(3 * n);
    hipMemcpy(r.data(), o, 3 * n * 4, hipMemcpyDeviceToHost);
    const char* names[3] = {"poly", "0.5-0.5*v_cos(x/pi)", "v_sin(x/2pi)^2"};
    for (int m = 0; m < 3; ++m) {
        double worst = 0, sum = 0;
        for (int i = 0; i < n; ++i) {
            const double ref = std::sin((double)h[i]); 
            const double e = std::fabs((double)r[m * n + i] - ref * ref);
            worst = e > worst ? e : worst; sum += e;
        }
        printf("%-22s max abs err %.3e  mean %.3e\n", names[m], worst, sum / n);
    }
    return 0;
}
