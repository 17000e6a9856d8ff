// Request dispatcher: coalesces concurrent single-utterance requests into batched forwards.
//
// The reference serves every request through one `Mutex<Session>` (kokorox/src/onn/ort_koko.rs:78; callers
// kokorox-openai/src/lib.rs:370-439, kokorox-websocket/src/lib.rs:657-668): N clients = N sequential runs.
// Here any number of OS threads call kx_dispatcher_submit(); one worker thread per model (= per GPU) takes
// up to `max_batch` queued requests — waiting at most `max_wait_us` for company once the first has arrived —
// and runs them as ONE kx_infer batch.  Every request carries its own noise seed, applied per utterance, so a
// request's waveform is bit-identical whether it ran alone or inside any batch (tests/test_gpu_dispatcher.py).
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstring>
#include <thread>

#include "../../include/kokorox_hip.h"
#include "kx_handle.h"

namespace {

struct Request {
    std::vector<int64_t> ids;
    std::vector<float> style;
    float speed = 1.f;
    uint64_t seed = 0;
    // result
    float* out = nullptr;
    int64_t out_len = 0;
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
    std::vector<std::thread> workers;
    int64_t n_requests = 0, n_batches = 0, max_seen_batch = 0;

    void worker(kx_model* h) {
        for (;;) {
            std::vector<Request*> batch;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop && queue.empty()) return;
                // first request is here: give others a short chance to join, unless the batch is already full
                const auto deadline = queue.front()->t_submit + std::chrono::microseconds(max_wait_us);
                while (!stop && (int)queue.size() < max_batch && std::chrono::steady_clock::now() < deadline)
                    cv_work.wait_until(lk, deadline);
                while (!queue.empty() && (int)batch.size() < max_batch) {
                    batch.push_back(queue.front());
                    queue.pop_front();
                }
                n_batches += 1;
                n_requests += (int64_t)batch.size();
                if ((int64_t)batch.size() > max_seen_batch) max_seen_batch = (int64_t)batch.size();
            }
            if (batch.empty()) continue;
            run_batch(h, batch);
            {
                std::lock_guard<std::mutex> lk(mu);
                for (Request* r : batch) r->done = true;
            }
            cv_done.notify_all();
        }
    }

    static void run_batch(kx_model* h, std::vector<Request*>& batch) {
        const int B = (int)batch.size();
        size_t stride = 0;
        for (Request* r : batch) stride = r->ids.size() > stride ? r->ids.size() : stride;
        std::vector<int64_t> ids((size_t)B * stride, 0);
        std::vector<int32_t> lens(B);
        std::vector<float> styles((size_t)B * KX_STYLE_DIM), speeds(B);
        std::vector<uint64_t> seeds(B);
        for (int b = 0; b < B; ++b) {
            lens[b] = (int32_t)batch[b]->ids.size();
            memcpy(&ids[(size_t)b * stride], batch[b]->ids.data(), batch[b]->ids.size() * 8);
            memcpy(&styles[(size_t)b * KX_STYLE_DIM], batch[b]->style.data(), KX_STYLE_DIM * 4);
            speeds[b] = batch[b]->speed;
            seeds[b] = batch[b]->seed;
        }
        float* out = nullptr;
        std::vector<int64_t> out_lens(B, 0);
        int rc = KX_ERR_DEVICE;
        std::string err;
        {
            kx::Model& M = *h->m;
            std::lock_guard<std::mutex> lk(M.mu);
            try {
                M.infer_host(ids.data(), (int64_t)stride, lens.data(), B, styles.data(), speeds.data(), B, 0, 0, &out,
                             out_lens.data(), seeds.data());
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
                r->out_len = out_lens[b];
                r->out = static_cast<float*>(malloc((size_t)(out_lens[b] > 0 ? out_lens[b] : 1) * sizeof(float)));
                if (r->out)
                    memcpy(r->out, out + off, (size_t)out_lens[b] * sizeof(float));
                else
                    r->rc = KX_ERR_DEVICE;
                off += out_lens[b];
            }
        }
        kx::host_out_free(out);  // (the batch buffer is one of the pooled page-locked ones)
    }
};

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
    for (kx_model* h : d->models) d->workers.emplace_back([d, h] { d->worker(h); });
    return d;
}

int kx_dispatcher_submit(kx_dispatcher* d, const int64_t* ids, int n_tokens, const float* style, float speed,
                         uint64_t seed, float** out, int64_t* out_len, char* err, size_t err_len) {
    if (!d || !ids || !style || !out || !out_len || n_tokens < 1 || n_tokens > KX_MAX_TOKENS || !(speed > 0.f)) {
        if (err && err_len) snprintf(err, err_len, "dispatcher_submit: bad argument (1..512 tokens, speed > 0)");
        return KX_ERR_INVALID;
    }
    for (int t = 0; t < n_tokens; ++t)
        if (ids[t] < 0 || ids[t] >= 178) {
            if (err && err_len) snprintf(err, err_len, "dispatcher_submit: token id outside 0..177");
            return KX_ERR_INVALID;
        }
    Request r;
    r.ids.assign(ids, ids + n_tokens);
    r.style.assign(style, style + KX_STYLE_DIM);
    r.speed = speed;
    r.seed = seed;
    r.t_submit = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> lk(d->mu);
        if (d->stop) {
            if (err && err_len) snprintf(err, err_len, "dispatcher is shutting down");
            return KX_ERR_STATE;
        }
        d->queue.push_back(&r);
        d->cv_work.notify_all();
        d->cv_done.wait(lk, [&] { return r.done; });
    }
    if (r.rc != KX_OK) {
        if (err && err_len) snprintf(err, err_len, "%s", r.err.c_str());
        free(r.out);
        return r.rc;
    }
    *out = r.out;
    *out_len = r.out_len;
    return KX_OK;
}

int kx_dispatcher_stats(kx_dispatcher* d, int64_t* n_requests, int64_t* n_batches, int64_t* max_batch_seen) {
    if (!d) return KX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(d->mu);
    if (n_requests) *n_requests = d->n_requests;
    if (n_batches) *n_batches = d->n_batches;
    if (max_batch_seen) *max_batch_seen = d->max_seen_batch;
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
