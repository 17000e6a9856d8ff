// Single-process multi-GPU host (BASELINE config 5 shape): one weight-file read, one model per visible GPU
// (at most 8; on a 1-GPU box two models share the card), a dispatcher over all of them, concurrent clients.
// Prints per-request FNV-1a checksums that tests/test_gpu_cpp_host.py compares with the ctypes path.
// Usage: replicas_demo <weights.kxw> <style.f32 (256 floats)> [n_models]
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <thread>

#include "kokorox_hip.hpp"

static uint64_t fnv1a(const float* p, int64_t n) {
    uint64_t sum = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t u;
        memcpy(&u, p + i, 4);
        for (int k = 0; k < 4; ++k) {
            sum ^= (u >> (8 * k)) & 0xff;
            sum *= 1099511628211ull;
        }
    }
    return sum;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s weights.kxw style.f32 [n_models]\n", argv[0]);
        return 2;
    }
    std::vector<float> style(256);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(style.data(), 4, 256, f) != 256) {
        fprintf(stderr, "cannot read style row\n");
        return 2;
    }
    fclose(f);
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) {
        fprintf(stderr, "no HIP device\n");
        return 1;
    }
    int n_models = argc > 3 ? atoi(argv[3]) : (visible < 8 ? visible : 8);
    if (n_models < 1) n_models = 1;
    std::vector<int> devices(n_models);
    for (int i = 0; i < n_models; ++i) devices[i] = i % visible;
    try {
        // (progress on stderr: should this demo ever stall, the test's time-out report says where)
        fprintf(stderr, "stage: creating %d replicas\n", n_models);
        std::vector<kokorox::HipKoko> models = kokorox::HipKoko::replicas(argv[1], devices);
        fprintf(stderr, "stage: replicas ready\n");
        std::vector<kx_model*> hs;
        for (auto& m : models) hs.push_back(m.handle());
        char err[256] = {0};
        kx_dispatcher* d = kx_dispatcher_create(hs.data(), (int)hs.size(), 8, 2000, err, sizeof(err));
        if (!d) throw std::runtime_error(err);
        const std::vector<int64_t> row = {0, 50, 83, 54, 156, 57, 135, 3, 16, 65, 156, 87, 158, 54, 46, 5, 0};
        const int n_req = 8;
        std::vector<uint64_t> sums(n_req, 0);
        std::vector<int64_t> lens(n_req, 0);
        std::vector<int> rcs(n_req, -1);
        std::vector<std::thread> th;
        for (int i = 0; i < n_req; ++i)
            th.emplace_back([&, i] {
                float* out = nullptr;
                char e2[256];
                // request i: the first 5 + i tokens of the row, wrapped in the pads again
                std::vector<int64_t> ids(row.begin(), row.begin() + 5 + i);
                ids.push_back(0);
                rcs[i] = kx_dispatcher_submit(d, ids.data(), (int)ids.size(), style.data(), 1.0f, 100 + i, &out, &lens[i], e2,
                                              sizeof(e2));
                if (rcs[i] == KX_OK) {
                    sums[i] = fnv1a(out, lens[i]);
                    kx_free_audio(out);
                }
            });
        for (auto& t : th) t.join();
        fprintf(stderr, "stage: requests answered\n");
        int64_t nr = 0, nb = 0, mb = 0;
        kx_dispatcher_stats(d, &nr, &nb, &mb);
        kx_dispatcher_destroy(d);
        fprintf(stderr, "stage: dispatcher destroyed\n");
        printf("models=%d requests=%lld\n", n_models, (long long)nr);
        for (int i = 0; i < n_req; ++i)
            printf("req%d rc=%d samples=%lld fnv1a=%016llx\n", i, rcs[i], (long long)lens[i], (unsigned long long)sums[i]);
        fflush(stdout);
        fprintf(stderr, "stage: destroying the models\n");
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    fprintf(stderr, "stage: done\n");
    return 0;
}
