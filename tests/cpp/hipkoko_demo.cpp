// Compiled-host check of the C ABI: run the reference's own token row (tokenize.rs:124-129) through
// kokorox::HipKoko and print sample count + a bit-exact checksum that tests/test_gpu_cpp_host.py compares
// with the ctypes path.  Usage: hipkoko_demo <weights.kxw> <style.f32 (256 floats)>
#include <cstdio>
#include <cstring>

#include "kokorox_hip.hpp"

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s weights.kxw style.f32\n", argv[0]);
        return 2;
    }
    std::vector<float> style(256);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(style.data(), 4, 256, f) != 256) {
        fprintf(stderr, "cannot read style row\n");
        return 2;
    }
    fclose(f);
    try {
        kokorox::HipKoko model(argv[1]);
        const std::vector<int64_t> row = {0, 50, 83, 54, 156, 57, 135, 3, 16, 65, 156, 87, 158, 54, 46, 5, 0};
        std::vector<int64_t> lens;
        std::vector<float> wav = model.infer({row}, {style}, 1.0f, /*seed=*/2, &lens);
        uint64_t sum = 1469598103934665603ull;  // FNV-1a over the raw bits
        for (float v : wav) {
            uint32_t u;
            memcpy(&u, &v, 4);
            for (int k = 0; k < 4; ++k) {
                sum ^= (u >> (8 * k)) & 0xff;
                sum *= 1099511628211ull;
            }
        }
        printf("samples=%lld fnv1a=%016llx\n", (long long)lens[0], (unsigned long long)sum);
        try {
            model.infer({}, {}, 1.0f);
        } catch (const std::invalid_argument&) {
            printf("empty-input=error\n");
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
