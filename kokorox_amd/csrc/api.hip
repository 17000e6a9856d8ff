// extern "C" surface of libkokorox_hip.so (include/kokorox_hip.h).  Nothing throws across it.
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>
#include <thread>
#include <chrono>
#include <mutex>

#include "../../include/kokorox_hip.h"
#include "model.h"

using kx::Error;
using kx::Model;

#include "kx_handle.h"

#include "api_guard.h"  // the exception fence + the per-thread last-error text (HIP-free: sanitizer-tested on the CPU)
using kx::guarded;
using kx::guarded_free;
using kx::set_err;

static void check_device(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) throw Error(KX_ERR_DEVICE, "no HIP device is visible (the HIP path has no CPU fallback)");
    if (device_id < 0 || device_id >= n) throw Error(KX_ERR_INVALID, "device id out of range");
    hipDeviceProp_t p;
    KX_HIP(hipGetDeviceProperties(&p, device_id));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        throw Error(KX_ERR_DEVICE, std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only");
}


extern "C" {

const char* kx_version(void) { return "kokorox-hip 0.4 (gfx950; conv modes: f16f8 split MFMA with 8-bit cross terms [default], f16x3 split MFMA, f32 MFMA, f16 / bf16 reduced precision [opt-in])"; }

int kx_init(int device_id, char* err, size_t err_len) {
    return guarded_free(err, err_len, [&] { check_device(device_id); });
}

kx_model* kx_create(const char* weights_path, int device_id, char* err, size_t err_len) {
    kx_model* h = nullptr;
    int rc = guarded_free(err, err_len, [&] {
        check_device(device_id);
        std::unique_ptr<Model> m(new Model(device_id));
        m->load_file(weights_path);
        h = new kx_model{std::move(m)};
    });
    return rc == KX_OK ? h : nullptr;
}

int kx_import_onnx(const char* onnx_path, const char* out_path, char* err, size_t err_len) {
    return guarded_free(err, err_len, [&] {
        KX_REQUIRE(onnx_path && *onnx_path && out_path && *out_path, "kx_import_onnx: two paths");
        FILE* f = fopen(onnx_path, "rb");
        if (!f) throw Error(KX_ERR_IO, std::string("cannot open weight file: ") + onnx_path);
        std::vector<unsigned char> in;
        unsigned char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + got);
        fclose(f);
        const std::vector<unsigned char> blob = kx::import_onnx_bytes(in.data(), in.size());
        FILE* o = fopen(out_path, "wb");
        if (!o) throw Error(KX_ERR_IO, std::string("cannot write ") + out_path);
        const bool ok = fwrite(blob.data(), 1, blob.size(), o) == blob.size();
        if (fclose(o) != 0 || !ok) throw Error(KX_ERR_IO, std::string("short write on ") + out_path);
    });
}

kx_model* kx_create_from_device_blob(const void* d_blob, size_t n_bytes, int device_id, char* err, size_t err_len) {
    kx_model* h = nullptr;
    int rc = guarded_free(err, err_len, [&] {
        check_device(device_id);
        std::unique_ptr<Model> m(new Model(device_id));
        m->load_device_blob(d_blob, n_bytes);
        h = new kx_model{std::move(m)};
    });
    return rc == KX_OK ? h : nullptr;
}

static std::mutex g_replica_times_mu;
static double g_replica_times[3] = {0, 0, 0};

int kx_replicas_times(double* out3) {
    if (!out3) return KX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_replica_times_mu);
    for (int i = 0; i < 3; ++i) out3[i] = g_replica_times[i];
    return KX_OK;
}

kx_model* kx_create_partition(const char* weights_path, int device_id, int part, int n_parts, char* err, size_t err_len) {
    kx_model* h = nullptr;
    int rc = guarded_free(err, err_len, [&] {
        check_device(device_id);
        std::unique_ptr<Model> m(new Model(device_id, part, n_parts));
        m->load_file(weights_path);
        h = new kx_model{std::move(m)};
    });
    return rc == KX_OK ? h : nullptr;
}

int kx_create_replicas(const char* weights_path, const int* device_ids, int n, kx_model** out_models, char* err,
                       size_t err_len) {
    if (out_models)
        for (int i = 0; i < n; ++i) out_models[i] = nullptr;
    std::vector<void*> blobs;       // per replica: its device blob until a model has adopted it
    std::vector<int> blob_dev;
    std::vector<hipStream_t> streams;
    std::vector<kx_model*> made;
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    double t_ms[3] = {0, 0, 0};  // file read (+ conversion) done, fan-out done, models built
    int rc = guarded_free(err, err_len, [&] {
        KX_REQUIRE(device_ids && out_models && n >= 1 && n <= 64, "create_replicas: 1..64 device ids and an output array");
        for (int i = 0; i < n; ++i) check_device(device_ids[i]);
        int variant = -2;
        const std::vector<unsigned char> host = kx::read_weight_file(weights_path, &variant);  // the ONE file read
        const size_t nb = host.size();
        t_ms[0] = ms_since(t0);
        blobs.assign(n, nullptr);
        blob_dev.assign(device_ids, device_ids + n);
        streams.assign(n, nullptr);
        // replica 0: host -> device over PCIe, once
        KX_HIP(hipSetDevice(device_ids[0]));
        KX_HIP(hipMalloc(&blobs[0], nb));
        KX_HIP(hipMemcpy(blobs[0], host.data(), nb, hipMemcpyHostToDevice));
        // every other replica: device -> device from replica 0, all copies in flight together (xGMI is
        // point-to-point: each destination has its own link to the source, so they do not share bandwidth)
        for (int i = 1; i < n; ++i) {
            KX_HIP(hipSetDevice(device_ids[i]));
            if (device_ids[i] != device_ids[0]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]) == hipSuccess && can) {
                    hipError_t pe = hipDeviceEnablePeerAccess(device_ids[0], 0);
                    if (pe != hipSuccess) (void)hipGetLastError();  // already enabled, or staged copies: both fine
                }
            }
            KX_HIP(hipMalloc(&blobs[i], nb));
            KX_HIP(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
            KX_HIP(hipMemcpyPeerAsync(blobs[i], device_ids[i], blobs[0], device_ids[0], nb, streams[i]));
        }
        for (int i = 1; i < n; ++i) {
            KX_HIP(hipSetDevice(device_ids[i]));
            KX_HIP(hipStreamSynchronize(streams[i]));
        }
        t_ms[1] = ms_since(t0);
        // A device id that is given k times gets k CU-partitioned models (each confined to 1 / k of the CUs: they run side by
        // side without competing for a CU -- Model::Model); KX_REPLICA_PARTITION=0: whole-device models that take turns.
        static const bool partition = !(getenv("KX_REPLICA_PARTITION") && atoi(getenv("KX_REPLICA_PARTITION")) == 0);
        std::vector<int> part(n, 0), parts(n, 1);
        if (partition)
            for (int i = 0; i < n; ++i) {
                int k = 0, j = 0;
                for (int q = 0; q < n; ++q) {
                    if (device_ids[q] != device_ids[i]) continue;
                    if (q < i) ++j;
                    ++k;
                }
                KX_REQUIRE(k <= 8, "create_replicas: at most 8 models per device");
                part[i] = j;
                parts[i] = k;
            }
        // the models are built side by side, one host thread each (weight repacking on the model's own GPU and stream)
        made.assign(n, nullptr);
        std::vector<std::string> build_err(n);
        std::vector<int> build_rc(n, KX_OK);
        std::vector<std::thread> th;
        for (int i = 0; i < n; ++i)
            th.emplace_back([&, i] {
                void* b = blobs[i];
                try {
                    std::unique_ptr<Model> m(new Model(device_ids[i], part[i], parts[i]));
                    blobs[i] = nullptr;  // the model owns it from here on (freed by its destructor, also on a failed build)
                    m->adopt_blob_guard(b);
                    if (getenv("KX_TEST_HOOKS") && getenv("KX_TEST_FAIL_REPLICA") && atoi(getenv("KX_TEST_FAIL_REPLICA")) == i)
                        throw Error(KX_ERR_IO, "create_replicas: injected failure of replica " + std::to_string(i) + " (KX_TEST_FAIL_REPLICA)");
                    m->load_device_blob(b, nb, /*adopt=*/true);
                    m->set_source_variant(variant);
                    made[i] = new kx_model{std::move(m)};
                } catch (const Error& e) {
                    build_rc[i] = e.code;
                    build_err[i] = e.what();
                } catch (const std::exception& e) {
                    build_rc[i] = KX_ERR_DEVICE;
                    build_err[i] = e.what();
                }
            });
        for (std::thread& t : th) t.join();
        t_ms[2] = ms_since(t0);
        for (int i = 0; i < n; ++i)
            if (build_rc[i] != KX_OK) throw Error(build_rc[i], "create_replicas: replica " + std::to_string(i) + ": " + build_err[i]);
    });
    {
        std::lock_guard<std::mutex> lk(g_replica_times_mu);
        for (int i = 0; i < 3; ++i) g_replica_times[i] = t_ms[i];
    }
    for (size_t i = 0; i < streams.size(); ++i)
        if (streams[i]) {
            (void)hipSetDevice(blob_dev[i]);
            (void)hipStreamDestroy(streams[i]);
        }
    for (size_t i = 0; i < blobs.size(); ++i)
        if (blobs[i]) {
            (void)hipSetDevice(blob_dev[i]);
            (void)hipFree(blobs[i]);
        }
    if (rc != KX_OK) {
        for (kx_model* h : made)
            if (h) kx_destroy(h);
        return rc;
    }
    for (int i = 0; i < n; ++i) out_models[i] = made[i];
    return KX_OK;
}

void kx_destroy(kx_model* m) {
    if (!m) return;
    try {
        delete m;
    } catch (...) {
    }
}

// The message is copied under the model's mutex into storage owned by the CALLING thread: another thread failing on
// the same model afterwards cannot change or free what this pointer refers to.
const char* kx_last_error(const kx_model* m) {
    if (!m || !m->m) return "null model";
    return kx::last_error_of(m);  // (the calling thread's own last failure on this model)
}

int kx_last_error_copy(const kx_model* m, char* buf, size_t buf_len) {
    if (!buf || !buf_len) return KX_ERR_INVALID;
    if (!m || !m->m) {
        set_err(buf, buf_len, "null model");
        return KX_ERR_INVALID;
    }
    set_err(buf, buf_len, kx::last_error_of(m));
    return KX_OK;
}

int kx_infer(kx_model* m, const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* styles,
             const float* speeds, int n_speed, uint64_t seed, uint32_t flags, float** out, int64_t* out_lens) {
    return guarded(m, [&](Model& M) { M.infer_host(ids, t_stride, lens, B, styles, speeds, n_speed, seed, flags, out, out_lens); });
}

void kx_free_audio(float* p) { kx::host_out_free(p); }
void kx_free_packed(void* p) { kx::host_out_free(p); }

int kx_set_voice_table(kx_model* m, const float* table, int n_voices) {
    return guarded(m, [&](Model& M) { M.set_voice_table(table, n_voices); });
}

int kx_infer_voices(kx_model* m, const int64_t* ids, int64_t t_stride, const int32_t* lens, int B,
                    const int32_t* voice_ids, const float* weights, int max_mix, const float* speeds, int n_speed,
                    uint64_t seed, uint32_t flags, int format, void** out, int64_t* out_bytes, int64_t* out_samples) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(voice_ids && weights, "infer_voices: null argument");
        Model::HostCall hc;
        hc.voice_ids = voice_ids;
        hc.weights = weights;
        hc.max_mix = max_mix;
        hc.format = format;
        M.infer_host_ex(ids, t_stride, lens, B, speeds, n_speed, seed, flags, hc, out, out_bytes, out_samples);
    });
}

int kx_infer_packed(kx_model* m, const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* styles,
                    const float* speeds, int n_speed, uint64_t seed, uint32_t flags, int format, void** out,
                    int64_t* out_bytes, int64_t* out_samples) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(styles, "infer_packed: null argument");
        Model::HostCall hc;
        hc.styles = styles;
        hc.format = format;
        M.infer_host_ex(ids, t_stride, lens, B, speeds, n_speed, seed, flags, hc, out, out_bytes, out_samples);
    });
}

int kx_infer_device(kx_model* m, const int64_t* d_ids, int64_t t_stride, const int32_t* lens_host, int B,
                    const float* d_styles, const float* speeds_host, int n_speed, uint64_t seed, uint32_t flags,
                    float* d_audio, int64_t audio_ld, int32_t* d_frames, int64_t* need_ld) {
    return guarded(m, [&](Model& M) {
        M.order_after_null_stream();  // (the caller's device inputs: typically written on the legacy null stream)
        try {
            M.infer_device(d_ids, t_stride, lens_host, B, d_styles, speeds_host, n_speed, seed, flags, d_audio, audio_ld,
                           d_frames, need_ld);
        } catch (const kx::LstmTimeout&) {
            // a token-axis recurrence timed out (seen at the forward's one host wait): the inputs are still where the caller put
            // them, so the call is simply made again, now on the streaming recurrence (same bits).  A time-out of the frame-axis
            // recurrence shows only at kx_sync, which then fails with KX_ERR_DEVICE: that call cannot be repeated from here.
            M.note_rerun();
            M.infer_device(d_ids, t_stride, lens_host, B, d_styles, speeds_host, n_speed, seed, flags, d_audio, audio_ld,
                           d_frames, need_ld);
        }
        M.order_null_stream_after();  // (the caller's later null-stream work waits, on the GPU, for the forward's end)
    });
}

int kx_sync(kx_model* m) {
    return guarded(m, [&](Model& M) {
        M.sync();
        M.note_clean_forward();
    });
}

int kx_model_info(kx_model* m, int64_t* out8) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(out8, "model_info: null argument");
        M.info(out8);
    });
}

int kx_model_status(kx_model* m, int64_t* out4) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(out4, "model_status: null argument");
        M.status(out4);
    });
}

int kx_set_pinned_durations(kx_model* m, const int32_t* pattern, int n) {
    return guarded(m, [&](Model& M) { M.set_pinned(pattern, n); });
}

int kx_warmup(kx_model* m, int B, int n_tokens, int frames_per_token) {
    return guarded(m, [&](Model& M) { M.warmup(B, n_tokens, frames_per_token); });
}

int kx_call_times(kx_model* m, double* out4) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(out4, "call_times: null argument");
        M.call_times(out4);
    });
}

int kx_arena_bytes(kx_model* m, int64_t* out3) {
    if (!out3) return KX_ERR_INVALID;
    return guarded(m, [&](Model& M) { M.arena_bytes(out3); });
}

int kx_set_conv_mode(kx_model* m, int mode) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(mode == kx::CONV_F32 || mode == kx::CONV_F16X3 || mode == kx::CONV_F16 || mode == kx::CONV_BF16 || mode == kx::CONV_F16F8,
                   "conv mode must be 0 (f32 MFMA), 1 (f16x3 split MFMA), 4 (f16, reduced precision), 5 (bf16, reduced precision) or 6 (f16f8: 8-bit cross terms)");
        M.set_conv_mode(mode);
    });
}

int kx_get_conv_mode(kx_model* m) { return (m && m->m) ? m->m->conv_mode : -1; }

int kx_set_stft_variant(kx_model* m, int variant) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(variant == kx::STFT_ONNX || variant == kx::STFT_TORCH,
                   "stft variant must be 0 (ONNX export: conv-based pair) or 1 (torch.stft / torch.istft)");
        M.sync();
        M.stft_variant = variant;
    });
}

int kx_get_stft_variant(kx_model* m) { return (m && m->m) ? m->m->stft_variant : -1; }

int kx_set_lanes(kx_model* m, int n_lanes) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(n_lanes >= 0 && n_lanes <= 4, "lanes must be 0 (by batch size) or 1..4");
        M.sync();
        M.set_lanes(n_lanes);
    });
}

int kx_set_utterance_base(kx_model* m, uint64_t utt_base) {
    return guarded(m, [&](Model& M) { M.utt_base = utt_base; });
}

int kx_profile_enable(kx_model* m, int on) {
    return guarded(m, [&](Model& M) { M.profile_enable(on != 0); });
}

int kx_profile_read(kx_model* m, int64_t* launches, double* total_ms, double* total_flops) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(launches && total_ms && total_flops, "profile_read: null argument");
        M.profile_read(launches, total_ms, total_flops);
    });
}

int kx_profile_detail(kx_model* m, double* out, int64_t cap_rows, int64_t* n_rows) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(n_rows, "profile_detail: null argument");
        *n_rows = (int64_t)M.prof_detail.size();
        if (!out) return;
        int64_t n = *n_rows < cap_rows ? *n_rows : cap_rows;
        for (int64_t i = 0; i < n; ++i) {
            const auto& r = M.prof_detail[i];
            double* o = out + i * 10;
            o[0] = r.rows; o[1] = r.Cin; o[2] = r.K; o[3] = r.dil; o[4] = r.stride; o[5] = r.store;
            o[6] = r.cols; o[7] = r.flops; o[8] = r.ms; o[9] = r.bytes;
        }
    });
}

int kx_profile_aux(kx_model* m, int64_t* stats_launches, double* stats_bytes) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(stats_launches && stats_bytes, "profile_aux: null argument");
        M.profile_aux(stats_launches, stats_bytes);
    });
}

int kx_diag_enable(kx_model* m, int on) {
    return guarded(m, [&](Model& M) { M.diag_enable(on != 0); });
}

int kx_diag_count(kx_model* m, int64_t* n) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(n, "diag_count: null argument");
        *n = (int64_t)M.diag_collect().size();
    });
}

int kx_diag_get(kx_model* m, int64_t i, char* name, size_t name_len, double* vals7) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(name && name_len && vals7, "diag_get: null argument");
        const auto& d = M.diag_collect();
        KX_REQUIRE(i >= 0 && i < (int64_t)d.size(), "diag_get: index out of range");
        set_err(name, name_len, d[i].name);
        vals7[0] = d[i].rows; vals7[1] = d[i].Cin; vals7[2] = d[i].K; vals7[3] = d[i].act_shift;
        vals7[4] = d[i].absmax; vals7[5] = d[i].rms; vals7[6] = d[i].count;
    });
}

int kx_set_act_prescale(kx_model* m, const char* conv_name, int log2_scale) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(conv_name, "set_act_prescale: null argument");
        M.set_act_shift(conv_name, log2_scale);
    });
}

int kx_debug_tap(kx_model* m, const char* name, int b, float* out, int64_t out_cap, int32_t* C, int32_t* L) {
    return guarded(m, [&](Model& M) {
        KX_REQUIRE(name && C && L, "debug_tap: null argument");
        const kx::Tap* t = M.find_tap(name);
        if (!t) throw Error(KX_ERR_STATE, std::string("no such tap (was KX_FLAG_TAPS set?): ") + name);
        KX_REQUIRE(b >= 0 && b < t->B, "debug_tap: utterance index out of range");
        *C = t->C;
        *L = t->L[b];
        if (!out) return;
        KX_REQUIRE(out_cap >= (int64_t)t->C * t->L[b], "debug_tap: output buffer too small");
        for (int c = 0; c < t->C; ++c)
            memcpy(out + (size_t)c * t->L[b], t->data.data() + ((size_t)b * t->C + c) * t->ld, (size_t)t->L[b] * 4);
    });
}

}  // extern "C"
