// `model.onnx` -> KXHIPW01 weight image, inside the library (plain C++17: no HIP, no protobuf package).
//
// The reference hands the path of the Hugging Face `.onnx` file to ONNX Runtime: `OrtKoko::new(model_path)`
// (kokorox/src/tts/koko.rs:570-573) -> `OrtBase::load_model` -> `commit_from_file`
// (kokorox/src/onn/ort_base.rs:14-39); the path is built at kokorox/src/utils/hf_cache.rs:128-158, with the seven
// variants of hf_cache.rs:135-144 (fp32, fp16, int8 x3, 4-bit x2).  `kx_create` receives that same path, so the
// library reads the file itself: a protobuf wire-format walk over ModelProto -> GraphProto -> {initializer, node},
// then the placement rules of kokorox_amd/importer.py (which stays as the tested Python mirror: both produce the same
// bytes, tests/test_onnx_cpp_cpu.py).
//
// This header is free of HIP so that the walker can also be built with g++ -fsanitize=address,undefined for the
// fuzz test of the CPU suite (tests/cpp/onnx_fuzz.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace kx {

// bumped whenever the placement rules change what the importer writes for the same .onnx (it stamps the opt-in cache: model.hip)
constexpr int KX_IMPORTER_VERSION = 1;

struct ImportError : std::runtime_error {
    explicit ImportError(const std::string& m) : std::runtime_error(m) {}
};

struct SpecEntry {
    std::string name;
    int ndim;
    int dims[4];
    size_t count() const {
        size_t c = 1;
        for (int i = 0; i < ndim; ++i) c *= (size_t)dims[i];
        return c;
    }
};

// The Kokoro-82M tensor table in blob order (the C++ twin of kokorox_amd/weights.py::tensor_spec()).
const std::vector<SpecEntry>& tensor_spec();

inline bool is_kxw_magic(const unsigned char* p, size_t n) {
    static const char m[8] = {'K', 'X', 'H', 'I', 'P', 'W', '0', '1'};
    if (n < 8) return false;
    for (int i = 0; i < 8; ++i)
        if (p[i] != (unsigned char)m[i]) return false;
    return true;
}

// ONNX ModelProto bytes -> KXHIPW01 image (byte-identical to what `python -m kokorox_amd.importer` writes).
// Throws ImportError: malformed / truncated protobuf, external-data tensors, unsupported tensor types, tensors of the
// table that could not be found (the message lists initialisers that were not placed), shape mismatches.
// What kind of file it was (hf_cache.rs:135-144 lists the reference's seven variants), from the graph's operators and the
// initialisers' types.  ONNX Runtime executes the quantised variants as DynamicQuantizeLinear -> MatMulInteger / ConvInteger /
// MatMulNBits, i.e. it also quantises the ACTIVATIONS at run time; this importer de-quantises the weights to f32 and the model
// then runs f32-class arithmetic -- NOT the reference's computation for those files (kx_model_info says so).
struct ImportInfo {
    int n_float = 0, n_half = 0, n_bfloat = 0, n_int8 = 0, n_int4_blocks = 0;  // initialisers by storage type (int4: MatMulNBits weights)
    int n_integer_ops = 0;      // MatMulInteger + ConvInteger + DynamicQuantizeLinear nodes
    int n_nbits_ops = 0;        // MatMulNBits nodes
    int n_dequant_ops = 0;      // DequantizeLinear nodes
    // 0 = the library's own container, 1 = fp32 ONNX, 2 = fp16 / bf16 ONNX (widened), 3 = 8-bit quantised ONNX (weights
    // de-quantised), 4 = 4-bit quantised ONNX (weights de-quantised)
    int variant() const {
        if (n_nbits_ops > 0 || n_int4_blocks > 0) return 4;
        if (n_integer_ops > 0 || n_dequant_ops > 0 || n_int8 > 0) return 3;
        if (n_half + n_bfloat > n_float) return 2;
        return 1;
    }
};
std::vector<unsigned char> onnx_to_kxw(const unsigned char* data, size_t n, ImportInfo* info = nullptr);

}  // namespace kx
