// C++ host-side mirror of the reference's `OrtKoko` (kokorox/src/onn/ort_koko.rs:13-91) over the C ABI
// in kokorox_hip.h.  Header-only; link with -lkokorox_hip.  Same two public operations, same meaning:
//   OrtKoko::new(model_path) -> Result<Self, String>          => HipKoko(model_path) (throws std::runtime_error)
//   OrtKoko::infer(tokens, styles, speed) -> Result<Array,..>  => infer(tokens, styles, speed) -> vector<float>
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "kokorox_hip.h"

namespace kokorox {

class HipKoko {
  public:
    explicit HipKoko(const std::string& model_path, int device = 0) {
        char err[512] = {0};
        h_ = kx_create(model_path.c_str(), device, err, sizeof(err));
        if (!h_) throw std::runtime_error(std::string("Failed to create Kokoro TTS model: ") + err);
    }
    // One model per device id from one read of the weight file (kx_create_replicas): what a single-process server
    // hands to kx_dispatcher_create (the reference holds one Arc<OrtKoko>, kokorox-openai/src/lib.rs:370-439).
    static std::vector<HipKoko> replicas(const std::string& model_path, const std::vector<int>& devices) {
        std::vector<kx_model*> hs(devices.size(), nullptr);
        char err[512] = {0};
        if (kx_create_replicas(model_path.c_str(), devices.data(), (int)devices.size(), hs.data(), err, sizeof(err)) != KX_OK)
            throw std::runtime_error(std::string("Failed to create Kokoro TTS model: ") + err);
        std::vector<HipKoko> out;
        out.reserve(hs.size());
        for (kx_model* h : hs) out.push_back(HipKoko(h));
        return out;
    }
    kx_model* handle() const { return h_; }
    HipKoko(const HipKoko&) = delete;
    HipKoko& operator=(const HipKoko&) = delete;
    HipKoko(HipKoko&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~HipKoko() { kx_destroy(h_); }

    // tokens: B rows already wrapped with the 0 pads (koko.rs:1169-1175); styles: B rows of 256 floats.
    // Returns the B waveforms back to back; lens (optional) receives the per-utterance sample counts.
    std::vector<float> infer(const std::vector<std::vector<int64_t>>& tokens,
                             const std::vector<std::vector<float>>& styles, float speed, uint64_t seed = 0,
                             std::vector<int64_t>* lens = nullptr) const {
        if (tokens.empty() || tokens[0].empty()) throw std::invalid_argument("infer: empty token list");
        if (styles.size() != tokens.size()) throw std::invalid_argument("infer: one style row per utterance");
        const int B = (int)tokens.size();
        size_t stride = 0;
        for (const auto& t : tokens) stride = t.size() > stride ? t.size() : stride;
        std::vector<int64_t> ids((size_t)B * stride, 0);
        std::vector<int32_t> tl(B);
        std::vector<float> st((size_t)B * KX_STYLE_DIM);
        for (int b = 0; b < B; ++b) {
            tl[b] = (int32_t)tokens[b].size();
            for (size_t i = 0; i < tokens[b].size(); ++i) ids[(size_t)b * stride + i] = tokens[b][i];
            if (styles[b].size() != KX_STYLE_DIM) throw std::invalid_argument("infer: style rows need 256 floats");
            for (int k = 0; k < KX_STYLE_DIM; ++k) st[(size_t)b * KX_STYLE_DIM + k] = styles[b][k];
        }
        float* out = nullptr;
        std::vector<int64_t> ol(B);
        const int rc = kx_infer(h_, ids.data(), (int64_t)stride, tl.data(), B, st.data(), &speed, 1, seed, 0, &out,
                                ol.data());
        if (rc != KX_OK) {
            char msg[512];
            kx_last_error_copy(h_, msg, sizeof(msg));  // (a copy of our own: other threads may fail on this model meanwhile)
            throw std::runtime_error(std::string("kokorox_hip error: ") + msg);
        }
        int64_t total = 0;
        for (int64_t v : ol) total += v;
        std::vector<float> wav(out, out + total);  // the same owned copy as ort_koko.rs:85
        kx_free_audio(out);
        if (lens) *lens = ol;
        return wav;
    }

  private:
    explicit HipKoko(kx_model* adopted) : h_(adopted) {}
    kx_model* h_ = nullptr;
};

}  // namespace kokorox
