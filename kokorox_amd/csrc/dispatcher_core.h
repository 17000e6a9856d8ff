// Request dispatcher, host logic only (queue, workers, shares, validation, error policy); HIP-free.  dispatcher.hip
// instantiates it on kx::Model, the CPU suite's sanitizer build on a stub backend (tests/cpp/host_sanitize.cpp, built with
// g++ -fsanitize=thread and -fsanitize=address).
//
// The reference serves every request through one `Mutex<Session>` (kokorox/src/onn/ort_koko.rs:78; callers
// kokorox-openai/src/lib.rs:370-439, kokorox-websocket/src/lib.rs:657-668): N clients = N sequential runs.
// Here any number of OS threads call submit*(); worker threads of each model (= each GPU) take requests off ONE shared
// queue and run them as one batched forward:
//   * a worker whose model is free and that finds work waits at most `max_wait_us` after the oldest request's arrival for
//     company (unless the queue already holds a full batch for every free model), then takes its SHARE of the queue:
//     ceil(queued / free models), at most `max_batch`.  32 requests waiting in front of 8 idle GPUs become 8 batches of 4, not one batch
//     of 32 on one GPU beside seven idle ones; a single worker that frees up while the others are busy takes up to a
//     whole batch (the reference's config 5: kokorox-openai, 32 clients over 8 GPUs).
//   * a request names its voice either as the 256-float style row (what `mix_styles` returns, koko.rs:1255-1306) or as
//     (voice id, weight) pairs into the device voice table, single voice or mix, and its output form (f32 mono / f32
//     stereo, koko.rs:1239-1246 / PCM16, kokorox-websocket/src/lib.rs:696-736); requests of every kind share one batch.
//   * every request carries its own noise seed, applied per utterance, so a request's bytes are identical whether it
//     ran alone or inside any batch on any of the models.
//   * a request is checked COMPLETELY when it is submitted (token ids, token count, speed, format, voice ids against the
//     smallest voice table of the models, mix size): a bad request is refused there and never reaches a batch, so one
//     client cannot make other clients' batch fail.
//   * if a batch still fails as a whole: an INVALID-class failure is re-run request by request and only the requests that
//     fail alone report it; a DEVICE-class failure (HIP error) is retried ONCE as a batch — never B times on a GPU that may
//     be faulted.  If it fails again the MODEL is marked failed (round 5): its workers stop taking requests, it no longer
//     counts as a free model, and the batch goes back to the head of the queue for the healthy models; a sticky fault on
//     one GPU of eight therefore costs its clients one retry, not a share of all traffic failing twice for ever.  Only when
//     no healthy model is left do requests report the failure (what is queued then, and every later submit).
//     kx_dispatcher_health / kx_dispatcher_failures show the state.
//   * a finished request wakes ITS client only (one condition variable per request, signalled under the queue's mutex):
//     with one shared condition variable every batch woke every waiting client of every model.
//
//   * two worker threads per model alternate, so that the host side of a finished batch (handing every client its bytes,
//     waking it) runs beside the NEXT batch's forward; a worker takes requests only when its model's GPU phase is free,
//     so what arrives during a forward accumulates into the next batch.
//
// Backend contract:
//   using Handle = ...;  using Out = ...;                                 one Handle per model; Out = a finished batch's packed result
//   static int forward(Handle*, std::vector<dispatch::Request*>&, Out&);  the batched forward: fills rc / err of EVERY request
//                                                                         (and Out on success), returns the batch's status
//   static void distribute(std::vector<dispatch::Request*>&, Out&);       gives every request its out / out_bytes / out_samples
//   static int n_voices(Handle*);                                         rows of the model's voice table (0 = none)
//   static int n_vocab(Handle*);                                          rows of its embedding tables (token ids are 0 .. n_vocab - 1)
//   static void free_out(void*);                                          releases a request's `out` (what distribute handed out)
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kokorox_hip.h"

namespace kx {
namespace dispatch {

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
    std::condition_variable cv;  // signalled (under the core's mutex) when `done` is set: wakes this request's client only
    std::chrono::steady_clock::time_point t_submit;
};

template <class Backend>
struct Core {
    using Handle = typename Backend::Handle;
    std::vector<Handle*> models;
    int max_batch = 64;
    int max_wait_us = 2000;
    std::mutex mu;
    std::condition_variable cv_work, cv_quiet;
    std::deque<Request*> queue;
    bool stop = false;
    static constexpr int WORKERS_PER_MODEL = 2;
    std::vector<char> busy;  // per model: a forward is running (its GPU phase); 0 = a worker may start the next batch
    std::vector<char> failed;  // per model: a batch failed with DEVICE twice in a row: the model takes no more work
    int64_t n_model_failures = 0, n_requeued = 0;
    int inside = 0;  // client threads inside submit() (shutdown waits for them before the object goes away)
    std::vector<std::thread> workers;
    int64_t n_requests = 0, n_batches = 0, max_seen_batch = 0, n_replayed = 0, n_retried = 0;
    std::vector<int64_t> per_model_batches;

    Core(Handle** ms, int n, int max_b, int wait_us) : models(ms, ms + n), max_batch(max_b), max_wait_us(wait_us) {
        per_model_batches.assign((size_t)n, 0);
        busy.assign((size_t)n, 0);
        failed.assign((size_t)n, 0);
        for (int i = 0; i < n * WORKERS_PER_MODEL; ++i) workers.emplace_back([this, i] { worker(i); });
    }
    Core(const Core&) = delete;
    Core& operator=(const Core&) = delete;

    // Queued requests are still served; submit() calls that arrive from now on are refused; returns when the workers have
    // drained the queue and every client thread has left submit().
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (std::thread& t : workers)
            if (t.joinable()) t.join();
        std::unique_lock<std::mutex> lk(mu);
        cv_quiet.wait(lk, [&] { return inside == 0; });
    }
    ~Core() { shutdown(); }

    // Two workers per model alternate: while one hands a finished batch's results to its clients (host work: copies, wake-ups),
    // the other already runs the next batch's forward.  `busy[m]` is the model's GPU phase: a worker takes requests off the
    // queue only when its model is free, so requests that arrive during a forward accumulate into the next batch.
    void worker(int wi) {
        const size_t m = (size_t)wi / WORKERS_PER_MODEL;
        Handle* h = models[m];
        for (;;) {
            std::vector<Request*> batch;
            bool more;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    // (after `stop` the queue is still drained: a worker leaves only when nothing is waiting)
                    cv_work.wait(lk, [&] { return failed[m] || (stop && queue.empty()) || (!queue.empty() && !busy[m]); });
                    if (failed[m]) return;  // (its sibling marked the model failed: nothing more runs on it)
                    if (queue.empty()) return;
                    // work is here and this model is free: give others a short chance to join, unless every free model
                    // already has a full batch waiting
                    bool go = true;
                    while (!stop && !failed[m] && !queue.empty() && !busy[m] && (long)queue.size() < (long)max_batch * free_models()) {
                        const auto deadline = queue.front()->t_submit + std::chrono::microseconds(max_wait_us);
                        const auto now = std::chrono::steady_clock::now();
                        if (now >= deadline) break;
#if defined(__SANITIZE_THREAD__)
                        // gcc 11's ThreadSanitizer does not intercept pthread_cond_clockwait (what a steady-clock wait becomes)
                        // and then reports the mutex as locked twice; a system-clock wait goes through pthread_cond_timedwait
                        cv_work.wait_until(lk, std::chrono::system_clock::now() + (deadline - now));
#else
                        cv_work.wait_until(lk, deadline);
#endif
                    }
                    if (failed[m]) return;
                    if (queue.empty() || busy[m]) go = false;  // (the sibling worker, or another model's, was quicker)
                    if (go) break;
                }
                // this model's share of what is waiting (the other free models' workers wake up on the same notify)
                const long fm = free_models();
                long take = ((long)queue.size() + fm - 1) / fm;
                take = take > max_batch ? max_batch : take;
                while (!queue.empty() && (long)batch.size() < take) {
                    batch.push_back(queue.front());
                    queue.pop_front();
                }
                busy[m] = true;
                n_batches += 1;
                per_model_batches[m] += 1;
                n_requests += (int64_t)batch.size();
                if ((int64_t)batch.size() > max_seen_batch) max_seen_batch = (int64_t)batch.size();
                more = !queue.empty();
            }
            if (more) cv_work.notify_all();  // (what is left is for the other free models)
            typename Backend::Out out;
            int rc = forward_retrying(h, batch, out);
            const bool replay = rc == KX_ERR_INVALID && batch.size() > 1;
            if (replay) {
                // per-request isolation: only the requests that fail alone report the failure
                for (Request* r : batch) {
                    std::vector<Request*> one{r};
                    typename Backend::Out o1;
                    if (forward_retrying(h, one, o1) == KX_OK) Backend::distribute(one, o1);
                }
            }
            bool model_failed = false, requeued = false;
            {
                std::lock_guard<std::mutex> lk(mu);
                busy[m] = false;  // the model's GPU phase is over: its other worker may start the next batch
                if (replay) n_replayed += (int64_t)batch.size();
                if (rc == KX_ERR_DEVICE) {
                    // failed, retried, failed again: the model is out.  Its batch goes back to the HEAD of the queue (in order)
                    // for the healthy models; with none left, it and everything queued report the failure.
                    model_failed = true;
                    if (!failed[m]) n_model_failures += 1;
                    failed[m] = 1;
                    if (healthy_models() > 0) {
                        for (auto it = batch.rbegin(); it != batch.rend(); ++it) queue.push_front(*it);
                        n_requeued += (int64_t)batch.size();
                        requeued = true;
                    } else {
                        for (Request* r : queue) {
                            r->rc = KX_ERR_DEVICE;
                            r->err = "dispatcher: no healthy model is left (" + batch[0]->err + ")";
                            finish(r);
                        }
                        queue.clear();
                    }
                }
            }
            cv_work.notify_all();
            if (rc == KX_OK) Backend::distribute(batch, out);  // host work, beside the next batch's forward
            if (!requeued) {
                std::lock_guard<std::mutex> lk(mu);
                for (Request* r : batch) finish(r);
            }
            if (model_failed) return;
        }
    }

    // (under mu) the request is complete: wake its client.  Signalled while the mutex is held: the client cannot leave submit()
    // -- and destroy the request with its condition variable -- before this thread has let go of the mutex.
    void finish(Request* r) {
        r->done = true;
        r->cv.notify_one();
    }

    long healthy_models() const {  // (under mu)
        long n = 0;
        for (char f : failed) n += f ? 0 : 1;
        return n;
    }
    long free_models() const {  // (under mu) at least 1: the caller's own model is free when this is asked
        long n = 0;
        for (size_t i = 0; i < busy.size(); ++i) n += (busy[i] || failed[i]) ? 0 : 1;
        return n > 0 ? n : 1;
    }

    // One forward; a DEVICE-class failure gets one more try as the same batch; a second failure marks the model failed (worker).
    int forward_retrying(Handle* h, std::vector<Request*>& batch, typename Backend::Out& out) {
        int rc = Backend::forward(h, batch, out);
        if (rc != KX_ERR_DEVICE) return rc;
        rc = Backend::forward(h, batch, out);
        std::lock_guard<std::mutex> lk(mu);
        n_retried += 1;
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
            if (healthy_models() == 0) {
                if (err && err_len) snprintf(err, err_len, "dispatcher: no healthy model is left (kx_dispatcher_health)");
                return KX_ERR_DEVICE;
            }
            ++inside;
            queue.push_back(&r);
            cv_work.notify_all();
            r.cv.wait(lk, [&] { return r.done; });
            if (--inside == 0) cv_quiet.notify_all();
        }
        if (r.rc != KX_OK) {
            if (err && err_len) snprintf(err, err_len, "%s", r.err.c_str());
            if (r.out) Backend::free_out(r.out);  // (never set on a failed request today; were it ever, it is not malloc'd memory)
            return r.rc;
        }
        *out = r.out;
        if (out_bytes) *out_bytes = r.out_bytes;
        if (out_samples) *out_samples = r.out_samples;
        return KX_OK;
    }

    // rows of the SMALLEST embedding table among the models: token ids must be valid on whichever model takes the request
    int vocab_everywhere() {
        int n = -1;
        for (Handle* h : models) {
            const int v = Backend::n_vocab(h);
            n = (n < 0 || v < n) ? v : n;
        }
        return n < 0 ? 0 : n;
    }

    bool check_common(const int64_t* ids, int n_tokens, float speed, int format, const char* who, char* err,
                      size_t err_len) {
        if (!ids || n_tokens < 1 || n_tokens > KX_MAX_TOKENS || !(speed > 0.f) || format < 0 || format > 2) {
            if (err && err_len) snprintf(err, err_len, "%s: bad argument (1..512 tokens, speed > 0, format 0..2)", who);
            return false;
        }
        const int nv = vocab_everywhere();  // (from the models' embedding tables: vocab.rs:5-20 has 178 rows, a checkpoint may differ)
        for (int t = 0; t < n_tokens; ++t)
            if (ids[t] < 0 || ids[t] >= nv) {
                if (err && err_len) snprintf(err, err_len, "%s: token id outside 0..%d", who, nv - 1);
                return false;
            }
        return true;
    }

    // rows of the SMALLEST voice table among the models (a request may land on any of them)
    int voices_everywhere() {
        int n = -1;
        for (Handle* h : models) {
            const int v = Backend::n_voices(h);
            n = (n < 0 || v < n) ? v : n;
        }
        return n < 0 ? 0 : n;
    }

    int submit_row(const int64_t* ids, int n_tokens, const float* style, float speed, uint64_t seed, float** out,
                   int64_t* out_len, char* err, size_t err_len) {
        if (!style || !out || !out_len) {
            if (err && err_len) snprintf(err, err_len, "dispatcher_submit: bad argument (1..512 tokens, speed > 0)");
            return KX_ERR_INVALID;
        }
        if (!check_common(ids, n_tokens, speed, 0, "dispatcher_submit", err, err_len)) return KX_ERR_INVALID;
        Request r;
        r.ids.assign(ids, ids + n_tokens);
        r.style.assign(style, style + KX_STYLE_DIM);
        r.speed = speed;
        r.seed = seed;
        void* p = nullptr;
        const int rc = submit(r, &p, nullptr, out_len, err, err_len);
        if (rc == KX_OK) *out = static_cast<float*>(p);
        return rc;
    }

    int submit_ex(const int64_t* ids, int n_tokens, const float* style, const int32_t* voice_ids, const float* weights,
                  int n_mix, float speed, uint64_t seed, int format, void** out, int64_t* out_bytes, int64_t* out_samples,
                  char* err, size_t err_len) {
        if (!out || !out_bytes || !out_samples) {
            if (err && err_len) snprintf(err, err_len, "dispatcher_submit_ex: null output argument");
            return KX_ERR_INVALID;
        }
        if (!check_common(ids, n_tokens, speed, format, "dispatcher_submit_ex", err, err_len)) return KX_ERR_INVALID;
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
            // ids against the voice tables NOW: inside a batch a bad id would fail every request it was batched with
            const int nv = voices_everywhere();
            bool any = false;
            for (int k = 0; k < n_mix; ++k) {
                if (voice_ids[k] >= nv) {
                    if (err && err_len)
                        snprintf(err, err_len, "dispatcher_submit_ex: voice id %d outside the table of %d voices%s", (int)voice_ids[k], nv,
                                 nv ? "" : " (kx_set_voice_table was not called on every model)");
                    return KX_ERR_INVALID;
                }
                any = any || voice_ids[k] >= 0;
            }
            if (!any || (!weights && voice_ids[0] < 0)) {  // (negative ids are skipped parts of a mix, koko.rs:1283)
                if (err && err_len) snprintf(err, err_len, "dispatcher_submit_ex: no voice given");
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
        return submit(r, out, out_bytes, out_samples, err, err_len);
    }
};

}  // namespace dispatch
}  // namespace kx
