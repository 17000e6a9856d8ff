// Request dispatcher: the C entry points and the kx::Model backend of the host logic in dispatcher_core.h (queue, workers,
// shares, submit-time validation, error policy — HIP-free there so that the CPU suite can run it under the thread and
// address sanitizers against a stub model).
//
// The reference serves every request through one `Mutex<Session>` (kokorox/src/onn/ort_koko.rs:78; callers
// kokorox-openai/src/lib.rs:370-439, kokorox-websocket/src/lib.rs:657-668): N clients = N sequential runs.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dispatcher_core.h"
#include "kx_handle.h"

namespace {

using kx::dispatch::Request;

// One batched forward on a model: requests of every kind / format in one HostCall (per-utterance kinds, formats, keys).
struct ModelBackend {
    using Handle = kx_model;
    struct Out {
        void* buf = nullptr;  // the packed batch in one pooled page-locked buffer (utterances back to back)
        std::vector<int64_t> bytes, samples;
    };
    static int n_voices(kx_model* h) { return h->m->n_voices(); }
    static int n_vocab(kx_model* h) { return h->m->n_vocab(); }
    static void free_out(void* p) { kx::host_out_free(p); }
    static int forward(kx_model* h, std::vector<Request*>& batch, Out& o) {
        const int B = (int)batch.size();
        size_t stride = 0;
        int mm = 1;
        bool any_voice = false, mixed_format = false;
        for (Request* r : batch) {
            stride = r->ids.size() > stride ? r->ids.size() : stride;
            mm = r->n_mix > mm ? r->n_mix : mm;
            any_voice = any_voice || r->kind != 0;
            mixed_format = mixed_format || r->format != batch[0]->format;
        }
        std::vector<int64_t> ids((size_t)B * stride, 0);
        std::vector<int32_t> lens(B), kinds(B), formats(B), vids((size_t)B * mm, -1);
        std::vector<float> styles((size_t)B * KX_STYLE_DIM, 0.f), speeds(B), weights((size_t)B * mm, 0.f);
        std::vector<uint64_t> seeds(B);
        for (int b = 0; b < B; ++b) {
            const Request& r = *batch[b];
            lens[b] = (int32_t)r.ids.size();
            memcpy(&ids[(size_t)b * stride], r.ids.data(), r.ids.size() * 8);
            kinds[b] = r.kind;
            formats[b] = r.format;
            if (r.kind == 0) memcpy(&styles[(size_t)b * KX_STYLE_DIM], r.style.data(), KX_STYLE_DIM * 4);
            for (int k = 0; k < r.n_mix; ++k) {
                vids[(size_t)b * mm + k] = r.voice_ids[k];
                weights[(size_t)b * mm + k] = r.weights[k];
            }
            speeds[b] = r.speed;
            seeds[b] = r.seed;
        }
        o.buf = nullptr;
        o.bytes.assign(B, 0);
        o.samples.assign(B, 0);
        int rc = KX_ERR_DEVICE;
        std::string err;
        {
            kx::Model& M = *h->m;
            std::lock_guard<std::mutex> lk(M.mu);
            try {
                kx::Model::HostCall hc;
                hc.utt_seeds = seeds.data();
                hc.format = batch[0]->format;
                if (mixed_format) hc.formats = formats.data();
                if (any_voice) {
                    hc.voice_ids = vids.data();
                    hc.weights = weights.data();
                    hc.max_mix = mm;
                    hc.styles = styles.data();  // (rows of the kind-0 requests; zeros elsewhere)
                    hc.kinds = kinds.data();
                } else {
                    hc.styles = styles.data();
                }
                M.infer_host_ex(ids.data(), (int64_t)stride, lens.data(), B, speeds.data(), B, 0, 0, hc, &o.buf, o.bytes.data(),
                                o.samples.data());
                rc = KX_OK;
            } catch (const kx::Error& e) {
                rc = e.code;
                err = e.what();
            } catch (const std::exception& e) {
                err = e.what();
            }
        }
        for (Request* r : batch) {
            r->rc = rc;
            r->err = err;
        }
        return rc;
    }
    // Every request gets a pointer INTO the batch's page-locked buffer (no per-request malloc + copy of a megabyte each: that
    // was ~5 ms per batch of host time); the buffer goes back to the pool when the last of them has been freed
    // (kx_free_audio / kx_free_packed -> host_out_free).
    static void distribute(std::vector<Request*>& batch, Out& o) {
        const int B = (int)batch.size();
        std::vector<void*> parts((size_t)B);
        int64_t off = 0;
        for (int b = 0; b < B; ++b) {
            parts[(size_t)b] = static_cast<char*>(o.buf) + off;
            off += o.bytes[(size_t)b];
        }
        // A client that keeps (caches, leaks) one result keeps its whole batch's buffer page-locked.  Beyond KX_PINNED_LIVE_CAP_MB
        // (default 4096) of such buffers still held, a batch's results are copied out into plain allocations instead -- the ~5 ms of
        // host time per batch the sharing saves are the price of clients that do not give results back -- and the buffer returns
        // to the pool at once.
        static const long cap_mb = getenv("KX_PINNED_LIVE_CAP_MB") ? atol(getenv("KX_PINNED_LIVE_CAP_MB")) : 4096;
        if (kx::host_out_live_bytes() + (size_t)off > (size_t)(cap_mb > 0 ? cap_mb : 0) * (size_t(1) << 20)) {
            std::vector<void*> copies;
            copies.reserve((size_t)B);
            bool ok = true;
            for (int b = 0; b < B && ok; ++b) {
                void* q = malloc((size_t)(o.bytes[(size_t)b] > 0 ? o.bytes[(size_t)b] : 1));
                ok = q != nullptr;
                if (ok) {
                    memcpy(q, parts[(size_t)b], (size_t)o.bytes[(size_t)b]);
                    copies.push_back(q);
                }
            }
            if (ok) {
                parts = copies;
                kx::host_out_free(o.buf);
            } else {  // (out of pageable memory: share after all)
                for (void* q : copies) free(q);
                kx::host_out_share(o.buf, parts.data(), B);
            }
        } else {
            kx::host_out_share(o.buf, parts.data(), B);
        }
        for (int b = 0; b < B; ++b) {
            Request* r = batch[(size_t)b];
            r->out = parts[(size_t)b];
            r->out_bytes = o.bytes[(size_t)b];
            r->out_samples = o.samples[(size_t)b];
        }
        o.buf = nullptr;
    }
};

}  // namespace

struct kx_dispatcher : kx::dispatch::Core<ModelBackend> {
    kx_dispatcher(kx_model** ms, int n, int max_b, int wait_us) : kx::dispatch::Core<ModelBackend>(ms, n, max_b, wait_us) {}
};

extern "C" {

kx_dispatcher* kx_dispatcher_create(kx_model** models, int n_models, int max_batch, int max_wait_us, char* err,
                                    size_t err_len) {
    if (!models || n_models < 1 || max_batch < 1 || max_batch > 4096 || max_wait_us < 0) {
        if (err && err_len) snprintf(err, err_len, "dispatcher: bad argument");
        return nullptr;
    }
    for (int i = 0; i < n_models; ++i)
        if (!models[i] || !models[i]->m) {
            if (err && err_len) snprintf(err, err_len, "dispatcher: null model");
            return nullptr;
        }
    try {
        return new kx_dispatcher(models, n_models, max_batch, max_wait_us);
    } catch (const std::exception& e) {
        if (err && err_len) snprintf(err, err_len, "dispatcher: %s", e.what());
        return nullptr;
    }
}

kx_dispatcher* kx_dispatcher_create_warm(kx_model** models, int n_models, int max_batch, int max_wait_us, int warm_tokens,
                                         int warm_frames_per_token, char* err, size_t err_len) {
    if (!models || n_models < 1 || max_batch < 1 || max_batch > 4096) {
        if (err && err_len) snprintf(err, err_len, "dispatcher: bad argument");
        return nullptr;
    }
    // one discarded forward of the largest shape the deployment expects on every model, the models side by side: arenas,
    // page-locked result buffers and tile tables exist before the first request (kx_warmup says why)
    std::vector<std::string> werr((size_t)n_models);
    std::vector<std::thread> th;
    for (int i = 0; i < n_models; ++i)
        th.emplace_back([&, i] {
            if (!models[i] || !models[i]->m) {
                werr[(size_t)i] = "null model";
                return;
            }
            kx::Model& M = *models[i]->m;
            std::lock_guard<std::mutex> lk(M.mu);
            try {
                M.warmup(max_batch, warm_tokens, warm_frames_per_token);
            } catch (const std::exception& e) {
                werr[(size_t)i] = e.what();
            }
        });
    for (std::thread& t : th) t.join();
    for (int i = 0; i < n_models; ++i)
        if (!werr[(size_t)i].empty()) {
            if (err && err_len) snprintf(err, err_len, "dispatcher: warm-up of model %d failed: %s", i, werr[(size_t)i].c_str());
            return nullptr;
        }
    return kx_dispatcher_create(models, n_models, max_batch, max_wait_us, err, err_len);
}

int kx_dispatcher_submit(kx_dispatcher* d, const int64_t* ids, int n_tokens, const float* style, float speed,
                         uint64_t seed, float** out, int64_t* out_len, char* err, size_t err_len) {
    if (!d) {
        if (err && err_len) snprintf(err, err_len, "dispatcher_submit: null dispatcher");
        return KX_ERR_INVALID;
    }
    return d->submit_row(ids, n_tokens, style, speed, seed, out, out_len, err, err_len);
}

int kx_dispatcher_submit_ex(kx_dispatcher* d, const int64_t* ids, int n_tokens, const float* style,
                            const int32_t* voice_ids, const float* weights, int n_mix, float speed, uint64_t seed,
                            int format, void** out, int64_t* out_bytes, int64_t* out_samples, char* err, size_t err_len) {
    if (!d) {
        if (err && err_len) snprintf(err, err_len, "dispatcher_submit_ex: null dispatcher");
        return KX_ERR_INVALID;
    }
    return d->submit_ex(ids, n_tokens, style, voice_ids, weights, n_mix, speed, seed, format, out, out_bytes, out_samples, err,
                        err_len);
}

int kx_dispatcher_stats(kx_dispatcher* d, int64_t* n_requests, int64_t* n_batches, int64_t* max_batch_seen) {
    if (!d) return KX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(d->mu);
    if (n_requests) *n_requests = d->n_requests;
    if (n_batches) *n_batches = d->n_batches;
    if (max_batch_seen) *max_batch_seen = d->max_seen_batch;
    return KX_OK;
}

int kx_dispatcher_failures(kx_dispatcher* d, int64_t* n_replayed, int64_t* n_retried) {
    if (!d) return KX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(d->mu);
    if (n_replayed) *n_replayed = d->n_replayed;
    if (n_retried) *n_retried = d->n_retried;
    return KX_OK;
}

int kx_dispatcher_health(kx_dispatcher* d, int32_t* healthy, int n_models, int64_t* n_model_failures, int64_t* n_requeued) {
    if (!d || n_models < 0 || (n_models > 0 && !healthy)) return KX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(d->mu);
    for (int i = 0; i < n_models && i < (int)d->failed.size(); ++i) healthy[i] = d->failed[(size_t)i] ? 0 : 1;
    if (n_model_failures) *n_model_failures = d->n_model_failures;
    if (n_requeued) *n_requeued = d->n_requeued;
    return KX_OK;
}

int kx_dispatcher_model_batches(kx_dispatcher* d, int64_t* per_model, int n_models) {
    if (!d || !per_model || n_models < 0) return KX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(d->mu);
    for (int i = 0; i < n_models && i < (int)d->per_model_batches.size(); ++i) per_model[i] = d->per_model_batches[i];
    return KX_OK;
}

void kx_dispatcher_destroy(kx_dispatcher* d) {
    if (!d) return;
    d->shutdown();  // serves what is queued, refuses new submits, waits for every client thread to leave
    delete d;
}

}  // extern "C"
