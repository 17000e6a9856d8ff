// Does hipExtStreamCreateWithCUMask work for an ordinary user on this pool, and which (XCC, CU) do the bits select?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(63508);  // HW_REG_XCC_ID
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    // keep the CU busy a little so that the grid spreads
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
static void run(const char* label, hipStream_t s) {
    const int N = 4096;
    unsigned* d;
    hipMalloc(&d, N * 8);
    hipLaunchKernelGGL(where, dim3(N), dim3(256), 0, s, d);
    hipError_t e = hipStreamSynchronize(s);
    std::vector<unsigned> h(N * 2);
    hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
    std::set<std::pair<unsigned, unsigned>> cus;
    std::set<unsigned> xccs;
    for (int i = 0; i < N; ++i) {
        const unsigned hw = h[i * 2], xcc = h[i * 2 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        cus.insert({xcc, (se << 5) | (sh << 4) | cu});
        xccs.insert(xcc);
    }
    printf("%s: sync %s; %zu distinct (xcc, se/sh/cu) pairs on %zu XCCs\n", label, hipGetErrorString(e), cus.size(), xccs.size());
    hipFree(d);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    hipStream_t s0;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    run("full mask", s0);
    for (int variant = 0; variant < 3; ++variant) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const char* label = variant == 0 ? "bits 0..127" : (variant == 1 ? "bits 128..255" : "even bits");
        for (int i = 0; i < 256; ++i) {
            const bool on = variant == 0 ? i < 128 : (variant == 1 ? i >= 128 : (i & 1) == 0);
            if (on) mask[i >> 5] |= 1u << (i & 31);
        }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        printf("hipExtStreamCreateWithCUMask(%s): %s\n", label, hipGetErrorString(e));
        if (e == hipSuccess) {
            run(label, s);
            hipStreamDestroy(s);
        }
    }
    return 0;
}
