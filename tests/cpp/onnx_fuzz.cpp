// Fuzz driver for the in-library ONNX reader (kokorox_amd/csrc/onnx_import.cpp), built by tests/test_onnx_cpp_cpu.py
// with g++ -fsanitize=address,undefined.  usage: onnx_fuzz <model.onnx> <n_mutations> <seed>
// Every mutated file is handed over in a heap buffer of EXACTLY its length, so a read past the end is an ASan report;
// the only acceptable outcomes are a converted blob or kx::ImportError.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "onnx_import.h"

static uint64_t rng_state;
static uint64_t rnd() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<unsigned char> orig;
    unsigned char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) orig.insert(orig.end(), buf, buf + got);
    fclose(f);
    const long n_mut = atol(argv[2]);
    rng_state = (uint64_t)atoll(argv[3]);
    long ok = 0, rejected = 0;
    for (long it = -1; it < n_mut; ++it) {
        size_t n = orig.size();
        const int kind = it < 0 ? -1 : (int)(rnd() % 7);
        if (kind == 0 && n) n = (size_t)(rnd() % n);                            // truncate anywhere
        if (kind == 1 && n) n = n - 1 - (size_t)(rnd() % (n < 64 ? n : 64));    // truncate near the end
        unsigned char* m = (unsigned char*)malloc(n ? n : 1);
        if (n) memcpy(m, orig.data(), n);
        // positions are drawn with a bias towards the front (graph nodes, tensor headers) and towards message borders
        auto pos = [&]() -> size_t {
            if (!n) return 0;
            const uint64_t r = rnd();
            return (r & 1) ? (size_t)((r >> 1) % (n < 65536 ? n : 65536)) : (size_t)((r >> 1) % n);
        };
        if (n) {
            if (kind == 2)
                for (int k = 0, c = 1 + (int)(rnd() % 8); k < c; ++k) m[pos()] ^= (unsigned char)(1u << (rnd() % 8));  // bit flips
            if (kind == 3) {  // a run of continuation bytes: varints that never end / huge lengths
                size_t p = pos();
                for (size_t k = 0, c = 1 + (size_t)(rnd() % 12); k < c && p + k < n; ++k) m[p + k] = 0xFF;
            }
            if (kind == 4)
                for (int k = 0, c = 1 + (int)(rnd() % 4); k < c; ++k) m[pos()] = (unsigned char)rnd();  // random bytes
            if (kind == 5) {  // copy one region over another (valid-looking fields in the wrong place)
                const size_t a = pos(), b = pos(), len = (size_t)(rnd() % 256);
                for (size_t k = 0; k < len && a + k < n && b + k < n; ++k) m[b + k] = m[a + k];
            }
            if (kind == 6) {  // small integers into varint positions: data types, dims, lengths
                const size_t p = pos();
                m[p] = (unsigned char)(rnd() % 24);
            }
        }
        try {
            const std::vector<unsigned char> blob = kx::onnx_to_kxw(m, n);
            if (!kx::is_kxw_magic(blob.data(), blob.size())) {
                printf("converted blob lacks the magic\n");
                return 1;
            }
            ++ok;
        } catch (const kx::ImportError&) {
            ++rejected;
        }
        free(m);
    }
    printf("%ld mutations + the original: %ld converted, %ld rejected\n", n_mut, ok, rejected);
    return 0;
}
