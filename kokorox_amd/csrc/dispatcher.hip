// Request dispatcher: coalesces concurrent single-utterance requests into batched forwards and spreads a waiting
// queue over the idle GPUs.
//
// The reference serves every request through one `Mutex<Session>` (kokorox/src/onn/ort_koko.rs:78; callers
// kokorox-openai/src/lib.rs:370-439, kokorox-websocket/src/lib.rs:657-668): N clients = N sequential runs.
// Here any number of OS threads call kx_dispatcher_submit*(); one worker thread per model (= per GPU) takes requests
// off ONE shared queue and runs them as one batched forward:
//   * a worker that finds work waits at most `max_wait_us` after the oldest request's arrival for company (unless the
//     queue already holds a full batch for every idle worker), then takes its SHARE of the queue: ceil(queued / idle
//     workers), at most `max_batch`.  32 requests waiting in front of 8 idle GPUs become 8 batches of 4, not one batch
//     of 32 on one GPU beside seven idle ones; a single worker that frees up while the others are busy takes up to a
//     whole batch (the reference's config 5: kokorox-openai, 32 clients over 8 GPUs).
//   * a request names its voice either as the 256-float style row (what `mix_styles` returns, koko.rs:1255-1306) or as
//     (voice id, weight) pairs into the device voice table (kx_set_voice_table on every model), single voice or mix,
//     and its output form (f32 mono / f32 stereo, koko.rs:1239-1246 / PCM16, kokorox-websocket/src/lib.rs:696-736);
//     requests of every kind share one batch (per-utterance kinds / formats in Model::HostCall).
//   * every request carries its own noise seed, applied per utterance, so a request's bytes are identical whether it
//     ran alone or inside any batch on any of the models (tests/test_gpu_dispatcher.py).
//   * errors are per request: if a batch fails as a whole, its requests are re-run one by one and only the ones that
//     fail alone report the error.
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstring>
#include <thread>

#include "../../include/kokorox_hip.h"
#include "kx_handle.h"

namespace {

constexpr int MAX_MIX = 16;

struct Request {
    std::vector<int64_t> ids;
    int kind = 0;  // 0 = style row, 1 = single voice, 2 = mix
    std::vector<float> style;
    int32_t voice_ids[MAX_MIX];
    float weights[MAX_MIX];
    int n_mix = 0;
    int format = 0;
    float speed = 1.f;
    uint64_t seed = 0;
    // result
    void* out = nullptr;
    int64_t out_bytes = 0, out_samples = 0;
    int rc = -1;
    std::string err;
    bool done = false;
    std::chrono::steady_clock::time_point t_submit;
};

}  // namespace

struct kx_dispatcher {
    std::vector<kx_model*> models;
    int max_batch = 64;
    int max_wait_us = 2000;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Request*> queue;
    bool stop = false;
    int idle = 0;  // workers waiting for work (or for company) right now
    std::vector<std::thread> workers;
    int64_t n_requests = 0, n_batches = 0, max_seen_batch = 0, n_retried = 0;
    std::vector<int64_t> per_model_batches;

    void worker(int wi) {
        kx_model* h = models[wi];
        for (;;) {
            std::vector<Request*> batch;
            {
                std::unique_lock<std::mutex> lk(mu);
                ++idle;
                cv_work.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop && queue.empty()) {
                    --idle;
                    return;
                }
                // work is here: give others a short chance to join, unless every idle worker already has a full batch
                while (!stop && !queue.empty() && (long)queue.size() < (long)max_batch * idle) {
                    const auto deadline = queue.front()->t_submit + std::chrono::microseconds(max_wait_us);
                    if (std::chrono::steady_clock::now() >= deadline) break;
                    cv_work.wait_until(lk, deadline);
                }
                // this worker's share of what is waiting (the other idle workers wake up on the same notify and take theirs)
                long take = ((long)queue.size() + idle - 1) / idle;
                take = take > max_batch ? max_batch : take;
                --idle;
                while (!queue.empty() && (long)batch.size() < take) {
                    batch.push_back(queue.front());
                    queue.pop_front();
                }
                if (batch.empty()) continue;  // (another worker was quicker)
                n_batches += 1;
                per_model_batches[wi] += 1;
                n_requests += (int64_t)batch.size();
                if ((int64_t)batch.size() > max_seen_batch) max_seen_batch = (int64_t)batch.size();
            }
            if (!queue_empty_hint()) cv_work.notify_all();  // (what is left is for the other idle workers)
            const int rc = run_batch(h, batch);
            if (rc != KX_OK && batch.size() > 1) {
                // per-request isolation: the batch failed as a whole; only the requests that fail alone report it
                {
                    std::lock_guard<std::mutex> lk(mu);
                    n_retried += (int64_t)batch.size();
                }
                for (Request* r : batch) {
                    std::vector<Request*> one{r};
                    run_batch(h, one);
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                for (Request* r : batch) r->done = true;
            }
            cv_done.notify_all();
        }
    }

    bool queue_empty_hint() {
        std::lock_guard<std::mutex> lk(mu);
        return queue.empty();
    }

    static int run_batch(kx_model* h, std::vector<Request*>& batch) {
        const int B = (int)batch.size();
        size_t stride = 0;
        int mm = 1;
        bool any_voice = false, any_style = false, mixed_format = false;
        for (Request* r : batch) {
            stride = r->ids.size() > stride ? r->ids.size() : stride;
            mm = r->n_mix > mm ? r->n_mix : mm;
            any_voice = any_voice || r->kind != 0;
            any_style = any_style || r->kind == 0;
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
        void* out = nullptr;
        std::vector<int64_t> out_bytes(B, 0), out_samples(B, 0);
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
                M.infer_host_ex(ids.data(), (int64_t)stride, lens.data(), B, speeds.data(), B, 0, 0, hc, &out, out_bytes.data(),
                                out_samples.data());
                rc = KX_OK;
            } catch (const kx::Error& e) {
                rc = e.code;
                err = e.what();
            } catch (const std::exception& e) {
                err = e.what();
            }
        }
        int64_t off = 0;
        for (int b = 0; b < B; ++b) {
            Request* r = batch[b];
            r->rc = rc;
            r->err = err;
            if (rc == KX_OK) {
                r->out_bytes = out_bytes[b];
                r->out_samples = out_samples[b];
                r->out = malloc((size_t)(out_bytes[b] > 0 ? out_bytes[b] : 1));
                if (r->out)
                    memcpy(r->out, static_cast<char*>(out) + off, (size_t)out_bytes[b]);
                else
                    r->rc = KX_ERR_DEVICE;
                off += out_bytes[b];
            }
        }
        kx::host_out_free(out);  // (the batch buffer is one of the pooled page-locked ones)
        return rc;
    }

    // common tail of the submit calls: queue the request, wait for its result
    int submit(Request& r, void** out, int64_t* out_bytes, int64_t* out_samples, char* err, size_t err_len) {
        r.t_submit = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(mu);
            if (stop) {
                if (err && err_len) snprintf(err, err_len, "dispatcher is shutting down");
                return KX_ERR_STATE;
            }
            queue.push_back(&r);
            cv_work.notify_all();
            cv_done.wait(lk, [&] { return r.done; });
        }
        if (r.rc != KX_OK) {
            if (err && err_len) snprintf(err, err_len, "%s", r.err.c_str());
            free(r.out);
            return r.rc;
        }
        *out = r.out;
        if (out_bytes) *out_bytes = r.out_bytes;
        if (out_samples) *out_samples = r.out_samples;
        return KX_OK;
    }
};

static bool check_common(kx_dispatcher* d, const int64_t* ids, int n_tokens, float speed, int format, const char* who,
                         char* err, size_t err_len) {
    if (!d || !ids || n_tokens < 1 || n_tokens > KX_MAX_TOKENS || !(speed > 0.f) || format < 0 || format > 2) {
        if (err && err_len) snprintf(err, err_len, "%s: bad argument (1..512 tokens, speed > 0, format 0..2)", who);
        return false;
    }
    for (int t = 0; t < n_tokens; ++t)
        if (ids[t] < 0 || ids[t] >= 178) {
            if (err && err_len) snprintf(err, err_len, "%s: token id outside 0..177", who);
            return false;
        }
    return true;
}

extern "C" {

kx_dispatcher* kx_dispatcher_create(kx_model** models, int n_models, int max_batch, int max_wait_us, char* err,
                                    size_t err_len) {
    if (!models || n_models < 1 || max_batch < 1 || max_batch > 4096 || max_wait_us < 0) {
        if (err && err_len) snprintf(err, err_len, "dispatcher: bad argument");
        return nullptr;
    }
    for (int i = 0; i < n_models; ++i)
        if (!models[i]) {
            if (err && err_len) snprintf(err, err_len, "dispatcher: null model");
            return nullptr;
        }
    kx_dispatcher* d = new kx_dispatcher();
    d->models.assign(models, models + n_models);
    d->max_batch = max_batch;
    d->max_wait_us = max_wait_us;
    d->per_model_batches.assign(n_models, 0);
    for (int i = 0; i < n_models; ++i) d->workers.emplace_back([d, i] { d->worker(i); });
    return d;
}

int kx_dispatcher_submit(kx_dispatcher* d, const int64_t* ids, int n_tokens, const float* style, float speed,
                         uint64_t seed, float** out, int64_t* out_len, char* err, size_t err_len) {
    if (!style || !out || !out_len) {
        if (err && err_len) snprintf(err, err_len, "dispatcher_submit: bad argument (1..512 tokens, speed > 0)");
        return KX_ERR_INVALID;
    }
    if (!check_common(d, ids, n_tokens, speed, 0, "dispatcher_submit", err, err_len)) return KX_ERR_INVALID;
    Request r;
    r.ids.assign(ids, ids + n_tokens);
    r.style.assign(style, style + KX_STYLE_DIM);
    r.speed = speed;
    r.seed = seed;
    void* p = nullptr;
    const int rc = d->submit(r, &p, nullptr, out_len, err, err_len);
    if (rc == KX_OK) *out = static_cast<float*>(p);
    return rc;
}

int kx_dispatcher_submit_ex(kx_dispatcher* d, const int64_t* ids, int n_tokens, const float* style,
                            const int32_t* voice_ids, const float* weights, int n_mix, float speed, uint64_t seed,
                            int format, void** out, int64_t* out_bytes, int64_t* out_samples, char* err, size_t err_len) {
    if (!out || !out_bytes || !out_samples) {
        if (err && err_len) snprintf(err, err_len, "dispatcher_submit_ex: null output argument");
        return KX_ERR_INVALID;
    }
    if (!check_common(d, ids, n_tokens, speed, format, "dispatcher_submit_ex", err, err_len)) return KX_ERR_INVALID;
    Request r;
    if (style) {
        if (voice_ids) {
            if (err && err_len) snprintf(err, err_len, "dispatcher_submit_ex: give the style row OR voice ids, not both");
            return KX_ERR_INVALID;
        }
        r.kind = 0;
        r.style.assign(style, style + KX_STYLE_DIM);
    } else {
        if (!voice_ids || n_mix < 1 || n_mix > MAX_MIX || n_tokens < 2 || (!weights && n_mix != 1)) {
            if (err && err_len)
                snprintf(err, err_len, "dispatcher_submit_ex: voices need 1..16 ids (one id when weights is null: a single "
                                       "voice) and the two 0 pads among the tokens");
            return KX_ERR_INVALID;
        }
        r.kind = weights ? 2 : 1;
        r.n_mix = n_mix;
        for (int k = 0; k < n_mix; ++k) {
            r.voice_ids[k] = voice_ids[k];
            r.weights[k] = weights ? weights[k] : 0.f;
        }
    }
    r.ids.assign(ids, ids + n_tokens);
    r.format = format;
    r.speed = speed;
    r.seed = seed;
    return d->submit(r, out, out_bytes, out_samples, err, err_len);
}

int kx_dispatcher_stats(kx_dispatcher* d, int64_t* n_requests, int64_t* n_batches, int64_t* max_batch_seen) {
    if (!d) return KX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(d->mu);
    if (n_requests) *n_requests = d->n_requests;
    if (n_batches) *n_batches = d->n_batches;
    if (max_batch_seen) *max_batch_seen = d->max_seen_batch;
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
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->stop = true;
    }
    d->cv_work.notify_all();
    for (std::thread& t : d->workers) t.join();
    delete d;
}

}  // extern "C"
