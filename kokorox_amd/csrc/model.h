// Host-side model object behind the C ABI (include/kokorox_hip.h).
#pragma once
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kx_common.h"

namespace kx {

struct TensorInfo {
    size_t offset = 0, nbytes = 0;
    int ndim = 0;
    int dims[4] = {0, 0, 0, 0};
};

struct ConvW {
    const float* w = nullptr;     // packed [co_tile][chunk][k][8][BM]
    const float* bias = nullptr;  // [Cout] or null
    int rows = 0, Cin = 0, K = 0, BM = 0, n_chunks = 0;
    int up_s = 0, up_cout = 0;  // polyphase transposed conv
    const void* w16 = nullptr;  // split-f16 image (conv_f16x3.hip)
    const void* w16b = nullptr; // bf16 form of it (built when KOKOROX_CONV=bf16 / kx_set_conv_mode(5) is first selected)
    const void* w8x = nullptr;  // 8-bit cross image of it (f16f8 mode: KOKOROX_CONV=f16f8 / kx_set_conv_mode(6); the S16 form's shapes only)
    int n_chunks16 = 0;
    float unscale = 1.f;        // 2^-ws
    std::string name;           // registry key (diagnostics, kx_set_act_prescale)
    int act_shift = 0;          // f16x3: the transformed input is multiplied by 2^act_shift before the hi/lo split
};

struct LstmW {
    ConvW ih;                     // rows 2048 = [fwd i f g o | rev i f g o], bias = b_ih + b_hh
    const float* whhT = nullptr;  // [2][256][1024]
};

// bump allocator over one device allocation; `measure` mode only counts
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    bool measure = false;
    void* alloc(size_t bytes) {
        const size_t a = (off + 255) & ~size_t(255);
        off = a + bytes;
        if (measure) return nullptr;
        if (off > cap) throw Error(3, "arena overflow (internal sizing bug)");
        return base + a;
    }
    float* f(size_t n) { return static_cast<float*>(alloc(n * sizeof(float))); }
    int* i(size_t n) { return static_cast<int*>(alloc(n * sizeof(int))); }
};

struct Tap {
    std::vector<float> data;  // [B][C][ld]
    int B = 0, C = 0, ld = 0;
    std::vector<int> L;
};

// activation view: [B][C][ld], utterance b valid for len.lens[b]*len.mul+len.add columns
struct T {
    float* p = nullptr;
    long bs = 0;
    int ld = 0;
    int C = 0;
    LenMap len{nullptr, 1, 0};
    int Lmax = 0;
    T rows(int r0, int n) const {
        T t = *this;
        t.p = p + (long)r0 * ld;
        t.C = n;
        return t;
    }
};

struct ConvOpts {
    const float* nmean = nullptr;
    const float* nscale = nullptr;
    const float* nshift = nullptr;
    int act = ACT_NONE;
    float slope = 0.f;
    const float* alpha = nullptr;
    int dil = 1, stride = 1, pad = 0, in_up2 = 0;
    const T* resid = nullptr;
    int accum = 0;
    float out_mul = 1.f, out_div = 1.f;
    int epi = EPI_NONE;
    int store = ST_NORMAL;
    int up_pad = 0, up_off = 0, up_reflect = 0;
    LenMap up_len{nullptr, 1, 0};  // ST_UPSCATTER: un-shifted output length
    float2* stat_part = nullptr;   // fuse InstanceNorm partial sums of the output into the epilogue
};

// the KXHIPW01 image behind `path`: the container itself (header checked), or built from the `.onnx` the reference passes
// *variant (optional): what the file was -- 0 KXHIPW01 container, 1 fp32 ONNX, 2 fp16 / bf16 ONNX, 3 8-bit quantised ONNX,
// 4 4-bit quantised ONNX (3, 4: weights de-quantised), -1 a cached conversion (KOKOROX_KXW_CACHE=1)
std::vector<unsigned char> read_weight_file(const char* path, int* variant = nullptr);
std::vector<unsigned char> import_onnx_bytes(const unsigned char* data, size_t n, int* variant = nullptr);  // ImportError -> Error(KX_ERR_IO)

// (KX_ERR_DEVICE class) a part of a resident-weights LSTM recurrence timed out waiting for its partner: the call is invalid,
// the model has switched to the streaming recurrence; host entry points re-run the call once
struct LstmTimeout : Error {
    explicit LstmTimeout(const std::string& m) : Error(3, m) {}
};

class Model {
  public:
    Model(int device, int part = 0, int n_parts = 1);  // n_parts > 1: confined to CUs [part, part + 1) x CUs / n_parts
    ~Model();
    void load_file(const char* path);
    void load_device_blob(const void* d_blob, size_t n, bool adopt = false);
    // take ownership of a hipMalloc'd blob BEFORE parsing it, so that it is freed with the model even when the
    // build throws (kx_create_replicas)
    void adopt_blob_guard(void* d_blob) { blob_ = static_cast<char*>(d_blob); }

    void infer_device(const int64_t* d_ids, int64_t t_stride, const int32_t* lens_host, int B,
                      const float* d_styles, const float* speeds_host, int n_speed, uint64_t seed, uint32_t flags,
                      float* d_audio, int64_t audio_ld, int32_t* d_frames, int64_t* need_ld);
    void infer_host(const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* styles,
                    const float* speeds, int n_speed, uint64_t seed, uint32_t flags, float** out,
                    int64_t* out_lens, const uint64_t* utt_seeds = nullptr);
    // general host entry behind kx_infer / kx_infer_voices / kx_infer_packed
    struct HostCall {
        const float* styles = nullptr;       // host [B][256], or
        const int32_t* voice_ids = nullptr;  // host [B][max_mix] into the device voice table
        const float* weights = nullptr;      // host [B][max_mix]
        int max_mix = 0;
        const uint64_t* utt_seeds = nullptr;
        int format = 0;                      // 0 f32 mono, 1 f32 stereo, 2 pcm16 mono
        // per-utterance forms (the dispatcher's mixed batches); null = the batch-wide fields above
        const int32_t* kinds = nullptr;      // host [B]: 0 = row of `styles`, 1 = single voice (copy), 2 = mix
        const int32_t* formats = nullptr;    // host [B]
    };
    void infer_host_ex(const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* speeds,
                       int n_speed, uint64_t seed, uint32_t flags, const HostCall& hc, void** out, int64_t* out_bytes,
                       int64_t* out_samples);
    // [0] recurrence in use: 0 = resident weights (two / four workgroups), 1 = the streaming fall-back; [1] hand-off time-outs so
    // far; [2] clean forwards left until the resident forms return (0 when they are in use); [3] calls re-run transparently
    void status(int64_t out[4]) const;
    // kx_infer_device + kx_sync: the forward finished cleanly (counts towards the resident forms' return)
    void note_clean_forward();
    void note_rerun() { n_rerun_ += 1; }
    static constexpr int LSTM_REARM_AFTER = 64;
    void set_voice_table(const float* table, int n_voices);
    int n_voices() const { return n_voices_.load(std::memory_order_acquire); }  // (read by the dispatcher without the mutex)
    int n_vocab() const { return n_vocab_; }                                    // (fixed once the model is built)
    void sync();
    void order_after_null_stream();
    void order_null_stream_after();
    void set_pinned(const int32_t* pattern, int n);
    // one forward of B utterances x n_tokens with every duration pinned to frames_per_token, output discarded: sizes the arenas,
    // the pooled page-locked buffer and the flat tile tables for that shape before the first real request
    void warmup(int B, int n_tokens, int frames_per_token);
    void arena_bytes(int64_t out[3]) const { out[0] = (int64_t)arenaT_.cap; out[1] = (int64_t)arenaF_.cap; out[2] = (int64_t)arenaIO_.cap; }
    // host-side milestones of the last infer_device call, ms from its entry: front half queued, front half done on the GPU (the
    // one host wait), back half planned, back half queued (= the call's return)
    void call_times(double out[4]) const { for (int i = 0; i < 4; ++i) out[i] = call_ms_[i]; }
    void profile_enable(bool on);
    void profile_read(int64_t* launches, double* ms, double* flops);
    struct ProfRec { int rows, Cin, K, dil, stride, store; double cols, flops; float ms; double bytes; };
    void profile_aux(int64_t* stats_launches, double* stats_bytes);  // unfused InstanceNorm statistics passes since the last read
    std::vector<ProfRec> prof_detail;  // filled by profile_read (one record per timed launch)
    // real-weights diagnostics: with diag on, every conv launch also measures its input (after the AdaIN affine)
    struct DiagRec { std::string name; int rows, Cin, K, act_shift; double absmax, rms, count; };
    void diag_enable(bool on);
    const std::vector<DiagRec>& diag_collect();  // syncs, converts the device slots of the last call(s)
    void set_act_shift(const std::string& conv_name, int shift);
    int get_act_shift(const std::string& conv_name) const;
    const Tap* find_tap(const std::string& name) const;

    int cu_partition(int* n_parts) const { *n_parts = n_parts_; return part_; }
    // [0] source variant (read_weight_file; -2 = built from a device blob: unknown), [1] 1 = the arithmetic is the reference's for
    // this file (fp32 / fp16 source or container), 0 = de-quantised weights run f32-class (NOT ORT's integer arithmetic),
    // [2] conv mode, [3] vocabulary rows, [4] voices in the device table, [5] CU partition, [6] partitions, [7] CUs of the model
    void info(int64_t out[8]) const;
    void set_source_variant(int v) { source_variant_ = v; }
    std::mutex mu;
    uint64_t utt_base = 0;
    void set_lanes(int n) { lanes_cfg_ = n < 0 ? 0 : (n > N_LANES ? N_LANES : n); }
    void set_conv_mode(int mode);  // (modes 5 / 6 build the bf16 / 8-bit cross weight images on first use)
    int conv_mode = CONV_F16F8;  // (default since round 5; KOKOROX_CONV=f16x3 / kx_set_conv_mode(1): three f16 MFMAs per product everywhere)
    int stft_variant = STFT_ONNX;  // the ONNX export's conv-based STFT pair (what the reference runs)
    int device;

  private:
    void build();  // parse table, pack weights, style-fc descriptors
    const float* wt(const std::string& name) const;
    const TensorInfo& info(const std::string& name) const;
    bool has(const std::string& name) const { return table_.count(name) != 0; }
    float* dev_alloc(size_t floats);
    void pack16(ConvW& c, const PackSrc& src, const float* wT, int n_src_floats[3]);
    ConvW make_conv(const std::string& name, bool bias = true);
    ConvW make_conv_cat(const std::vector<std::string>& names);
    ConvW make_convT(const std::string& name, int stride);
    LstmW make_lstm(const std::string& name);
    void add_fc(const std::string& key, const std::string& fc_name, int style_off);
    long fc_off(const std::string& key) const;

    void conv(const ConvW& w, const T& in, const T& out, const ConvOpts& o);
    void stats(const T& x, const std::string& fc_key);
    void adain_resblk(const std::string& name, const T& x, const T& out, bool upsample, float* ws_a, float* ws_b,
                      float* ws_c);
    void adain_resblock1(const std::string& name, int k, const T& x, const T& xj, const T& t1, const T& out,
                         int accum, float out_div, float2* part_t1, float2* part_xj, hipEvent_t wait_before_last = nullptr);
    void lstm(const LstmW& w, const T& in, const T& out, float* gx);
    void tap(const char* name, const T& t);
    void ensure_arena(Arena& a, size_t bytes);

    // Lanes of the back half: chains that do not depend on each other (the harmonic-source / noise path, the three
    // resblocks of a generator stage) are issued on streams of their own and meet through events, so that the HBM-bound
    // phases of one chain (epilogues, statistics passes, k = 3 convs) run under the matrix work of another.  Lane 0 is the
    // main stream.  Every lane has its own InstanceNorm parameter set (stats() writes it, the next conv reads it).
    static constexpr int N_LANES = 4;
    struct Lane {
        hipStream_t stream = nullptr;
        float *nmean = nullptr, *nscale = nullptr, *nshift = nullptr;
    };
    Lane lanes_[N_LANES];
    int lanes_cfg_ = 0;  // 0 = by batch size (4 up to 32 utterances, else 1), 1..4 = fixed (KX_LANES, kx_set_lanes)
    int n_lanes_ = 1;    // lanes of the running call
    std::vector<hipEvent_t> lane_ev_;
    size_t lane_ev_used_ = 0;
    hipEvent_t record_here();               // a pooled event recorded on the current stream (null in the sizing pass)
    void wait_here(hipEvent_t e);           // the current stream waits for it
    struct LaneScope;                       // issue on lane k until the scope ends (model.hip)
    struct DeviceTurn;                      // one forward at a time per GPU across models (model.hip)
    void sync_lanes();

    int part_ = 0, n_parts_ = 1, cu_count_ = 0;  // cu_count_ = 0: the whole device
    int source_variant_ = -2;
    std::vector<uint32_t> cu_mask_;
    void new_stream(hipStream_t* s);
    hipStream_t stream_ = nullptr;
    // side stream for the TextEncoder branch, which does not depend on the ALBERT / duration branch
    hipStream_t stream2_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    char* blob_ = nullptr;
    size_t blob_bytes_ = 0;
    std::map<std::string, TensorInfo> table_;
    std::vector<void*> owned_;
    std::map<std::string, ConvW> convs_;
    std::map<std::string, LstmW> lstms_;
    std::vector<FcDesc> fc_host_;
    std::map<std::string, long> fc_off_;
    FcDesc* fc_dev_ = nullptr;
    long gb_total_ = 0;
    int n_vocab_ = 178;  // rows of the two embedding tables (vocab.rs:5-20)

    Arena arenaT_, arenaF_, arenaIO_;
    const uint64_t* d_utt_seeds_ = nullptr;  // per-utterance noise keys of the running call (dispatcher)
    float* d_voices_ = nullptr;  // [n_voices_][511][256]
    std::atomic<int> n_voices_{0};
    int* d_pinned_ = nullptr;
    Arena* stats_arena_ = nullptr;  // where stats() keeps the raw sums of tensors it had to read (frame-axis arena)
    Arena* img_arena_ = nullptr;    // where conv() puts pre-split input images; non-null only while the back half is issued
    int n_pinned_ = 0;

    // per-call state
    int B_ = 0, Tmax_ = 0, Fmax_ = 0;
    std::vector<int> hT_, hF_;
    int *dT_ = nullptr, *dF_ = nullptr;
    // two-CU LSTM: exchange buffers (one for the main stream, one for the TextEncoder branch that runs beside it) and the
    // sticky device error word (bit 1: a half never saw its partner) checked at every host synchronisation
    unsigned long long* d_xchg_[2] = {nullptr, nullptr};
    size_t xchg_cap_ = 0;
    unsigned xchg_epoch_[2] = {0, 0};  // launches on each exchange buffer since it was last cleared (16-bit tag epoch)
    bool p1_region_ = false;           // inside the part of the forward whose direct-A convs may run reduced precision
    bool lstm_pair_ok_ = true;         // false after a hand-off time-out: the streaming recurrence until lstm_rearm_in_ forwards were clean
    int lstm_rearm_in_ = 0;
    int64_t n_lstm_timeouts_ = 0, n_rerun_ = 0;
    void infer_host_once(const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* speeds, int n_speed,
                         uint64_t seed, uint32_t flags, const HostCall& hc, void** out, int64_t* out_bytes, int64_t* out_samples);
    std::vector<long> h_off_;          // host staging that asynchronous copies read / write: outlives the calling frame
    unsigned h_bad_id_ = 0;
    unsigned* d_dev_err_ = nullptr;
    unsigned* h_words_ = nullptr;    // page-locked: [0] the device error word as read back
    int* h_stage_ = nullptr;         // page-locked scratch of the small per-call host arrays (stage_ints)
    size_t h_stage_cap_ = 0;
    int* stage_ints(size_t n);
    hipEvent_t ev_null_ = nullptr;   // orders kx_infer_device's inputs after the caller's null-stream work
    hipEvent_t ev_done_ = nullptr;   // ... and the caller's later null-stream work after the forward (order_null_stream_after)
    hipStream_t main_stream_ = nullptr;
    void check_dev_err();
    unsigned* d_bad_id_ = nullptr;  // sticky word: first out-of-table token id seen by the embedding kernels
    float* gb_ = nullptr;  // [B][gb_total_]
    float *nmean_ = nullptr, *nscale_ = nullptr, *nshift_ = nullptr;
    int n_bs_ = 0;
    bool taps_on_ = false;
    bool dry_ = false;  // sizing pass: allocate (count) but launch nothing
    std::map<std::string, Tap> taps_;

    // flat tile lists of the running call (ConvArgs::tile_prefix), one per (length map, tile width, stream)
    struct PrefixKey { const int* lens; int mul, add, bn; hipStream_t stream; const int* dev; };
    std::vector<PrefixKey> prefix_keys_;
    int* d_prefix_ = nullptr;
    size_t prefix_cap_ = 0, prefix_used_ = 0;
    const int* tile_prefix_for(const LenMap& lm, int extra, int bn, int* total);

    struct PartInfo { const float2* part; int tiles, cols_per_tile, C; };
    std::map<const float*, PartInfo> parts_;  // output tensor -> fused statistics partials of its producer

    double call_ms_[4] = {0, 0, 0, 0};
    bool diag_on_ = false;
    std::vector<DiagRec> diag_recs_;
    float* d_diag_ = nullptr;  // [diag_cap_][3]
    size_t diag_cap_ = 0, diag_used_ = 0;
    bool prof_on_ = false;
    std::vector<hipEvent_t> ev_;
    size_t ev_used_ = 0;
    double prof_flops_ = 0.0;
    std::vector<ProfRec> prof_recs_;
    int64_t prof_launches_ = 0;
    int64_t prof_stats_launches_ = 0;
    double prof_stats_bytes_ = 0.0;
};

}  // namespace kx
