// Sanitizer build of the library's host-side concurrency (no GPU, no HIP): the dispatcher's queue / workers / shutdown
// (kokorox_amd/csrc/dispatcher_core.h) and the C ABI's exception fence + per-thread last-error text (api_guard.h), against a
// stub model.  Built and run by tests/test_host_sanitize_cpu.py with g++ -fsanitize=thread and -fsanitize=address,undefined.
//
// What the reference hand-asserts (`unsafe impl Send / Sync for OrtKoko` over a `Mutex<Session>`,
// kokorox/src/onn/ort_koko.rs:14,17-18,78) is checked here by the tools: 64 client threads, mixed voices and formats,
// batches that fail as a whole (INVALID: replayed one by one; DEVICE: retried once), a model that stays broken, requests
// refused at submit, and destroy-while-queued.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "api_guard.h"
#include "dispatcher_core.h"

using kx::dispatch::Request;

namespace {

struct StubModel {
    std::mutex mu;                  // calls on one model are serialised, as kx::Model::mu
    std::atomic<int> voices{54};
    std::atomic<long> batches{0};
    int fail_every = 0;             // every n-th batch fails once with DEVICE (the retry then succeeds)
    bool dead = false;              // every batch fails with DEVICE
    int sleep_us = 300;
    int distribute_us = 150;
    int vocab = 178;
};
struct StubHandle {
    std::unique_ptr<StubModel> m;
};

constexpr float POISON_SPEED = 13.0f;  // a request the submit-time checks cannot refuse, but that fails any batch it is in

uint32_t mix(uint32_t h, uint32_t v) { return (h ^ v) * 16777619u; }
int bps(int format) { return format == 1 ? 8 : (format == 2 ? 2 : 4); }

// what a request's bytes must be, whatever it was batched with
void expected(const Request& r, std::vector<unsigned char>& out) {
    uint32_t h = 2166136261u;
    for (int64_t id : r.ids) h = mix(h, (uint32_t)id);
    h = mix(h, (uint32_t)r.seed);
    h = mix(h, (uint32_t)r.kind * 31u + (uint32_t)r.format);
    for (int k = 0; k < r.n_mix; ++k) h = mix(h, (uint32_t)r.voice_ids[k] + 7u);
    out.resize(r.ids.size() * 10 * (size_t)bps(r.format));
    for (size_t i = 0; i < out.size(); ++i) out[i] = (unsigned char)((h >> ((i & 3) * 8)) + i);
}

struct StubBackend {
    using Handle = StubHandle;
    struct Out {
        std::vector<std::vector<unsigned char>> parts;  // the "packed batch": one entry per request
        int distribute_us = 0;
    };
    static int n_voices(StubHandle* h) { return h->m->voices.load(); }
    static int n_vocab(StubHandle* h) { return h->m->vocab; }
    static void free_out(void* p) { free(p); }
    static int forward(StubHandle* h, std::vector<Request*>& batch, Out& o) {
        StubModel& M = *h->m;
        std::unique_lock<std::mutex> lk(M.mu, std::try_to_lock);
        if (!lk.owns_lock()) {  // the dispatcher's busy flag must have kept the sibling worker out
            for (Request* r : batch) {
                r->rc = KX_ERR_STATE;
                r->err = "stub: two forwards on one model at once";
            }
            return KX_ERR_STATE;
        }
        const long nth = ++M.batches;
        if (M.sleep_us) std::this_thread::sleep_for(std::chrono::microseconds(M.sleep_us));
        int rc = KX_OK;
        std::string err;
        if (M.dead || (M.fail_every && nth % M.fail_every == 0)) {
            rc = KX_ERR_DEVICE;
            err = "stub: device failure";
        } else {
            for (Request* r : batch)
                if (r->speed == POISON_SPEED) {
                    rc = KX_ERR_INVALID;
                    err = "stub: poison request";
                }
        }
        o.parts.clear();
        o.distribute_us = M.distribute_us;
        for (Request* r : batch) {
            r->rc = rc;
            r->err = err;
            if (rc == KX_OK) {
                o.parts.emplace_back();
                expected(*r, o.parts.back());
            }
        }
        return rc;
    }
    static void distribute(std::vector<Request*>& batch, Out& o) {
        if (o.distribute_us) std::this_thread::sleep_for(std::chrono::microseconds(o.distribute_us));  // host work beside the next forward
        for (size_t i = 0; i < batch.size(); ++i) {
            Request* r = batch[i];
            const std::vector<unsigned char>& e = o.parts[i];
            r->out = malloc(e.size() ? e.size() : 1);
            memcpy(r->out, e.data(), e.size());
            r->out_bytes = (int64_t)e.size();
            r->out_samples = (int64_t)r->ids.size() * 10;
        }
    }
};

using Core = kx::dispatch::Core<StubBackend>;

int failures = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                 \
            fprintf(stderr, "\n");                        \
            ++failures;                                   \
        }                                                 \
    } while (0)

struct ClientTotals {
    std::atomic<long> ok{0}, invalid{0}, device{0}, state{0}, wrong{0};
};

// one request through submit_ex / submit_row; compares the bytes
int one_request(Core& core, unsigned& rng, bool poison, ClientTotals& tot) {
    auto rnd = [&] { return rng = rng * 1664525u + 1013904223u, rng >> 8; };
    const int n_tok = 2 + (int)(rnd() % 40);
    std::vector<int64_t> ids((size_t)n_tok);
    for (auto& v : ids) v = (int64_t)(rnd() % 178);
    Request ref;
    ref.ids = ids;
    ref.seed = rnd();
    ref.format = (int)(rnd() % 3);
    ref.kind = (int)(rnd() % 3);
    const float speed = poison ? POISON_SPEED : 1.0f;
    std::vector<float> style(KX_STYLE_DIM, 0.25f);
    int32_t vids[4] = {(int32_t)(rnd() % 54), -1, (int32_t)(rnd() % 54), (int32_t)(rnd() % 54)};
    float w[4] = {0.4f, 0.f, 0.5f, 0.1f};
    char err[256] = {0};
    void* out = nullptr;
    int64_t nb = 0, ns = 0;
    int rc;
    if (ref.kind == 0 && ref.format == 0 && (rnd() & 1)) {
        float* fo = nullptr;
        rc = core.submit_row(ids.data(), n_tok, style.data(), speed, ref.seed, &fo, &ns, err, sizeof err);
        out = fo;
        nb = ns * 4;
    } else if (ref.kind == 0) {
        rc = core.submit_ex(ids.data(), n_tok, style.data(), nullptr, nullptr, 0, speed, ref.seed, ref.format, &out, &nb, &ns, err, sizeof err);
    } else if (ref.kind == 1) {
        ref.n_mix = 1;
        ref.voice_ids[0] = vids[0];
        rc = core.submit_ex(ids.data(), n_tok, nullptr, vids, nullptr, 1, speed, ref.seed, ref.format, &out, &nb, &ns, err, sizeof err);
    } else {
        ref.n_mix = 4;
        for (int k = 0; k < 4; ++k) ref.voice_ids[k] = vids[k];
        rc = core.submit_ex(ids.data(), n_tok, nullptr, vids, w, 4, speed, ref.seed, ref.format, &out, &nb, &ns, err, sizeof err);
    }
    if (rc == KX_OK) {
        std::vector<unsigned char> e;
        expected(ref, e);
        if ((int64_t)e.size() != nb || memcmp(e.data(), out, e.size()) != 0 || ns != (int64_t)n_tok * 10) ++tot.wrong;
        free(out);
        ++tot.ok;
        if (poison) ++tot.wrong;
    } else if (rc == KX_ERR_INVALID) {
        ++tot.invalid;
        if (!poison || !strstr(err, "poison")) ++tot.wrong;
    } else if (rc == KX_ERR_DEVICE) {
        ++tot.device;
    } else if (rc == KX_ERR_STATE) {
        ++tot.state;
    } else {
        ++tot.wrong;
    }
    return rc;
}

std::vector<StubHandle> make_models(int n) {
    std::vector<StubHandle> hs((size_t)n);
    for (auto& h : hs) h.m.reset(new StubModel);
    return hs;
}

void scenario_mixed_load() {
    std::vector<StubHandle> hs = make_models(3);
    hs[1].m->fail_every = 7;  // one model has transient device failures: every request must still succeed
    StubHandle* ptr[3] = {&hs[0], &hs[1], &hs[2]};
    ClientTotals tot;
    {
        Core core(ptr, 3, 16, 500);
        std::vector<std::thread> th;
        for (int c = 0; c < 64; ++c)
            th.emplace_back([&, c] {
                unsigned rng = 1234u + (unsigned)c * 977u;
                for (int i = 0; i < 30; ++i) one_request(core, rng, /*poison=*/(c * 30 + i) % 19 == 0, tot);
            });
        for (auto& t : th) t.join();
        std::lock_guard<std::mutex> lk(core.mu);
        CHECK(core.n_requests == 64 * 30, "n_requests %ld", (long)core.n_requests);
        CHECK(core.n_replayed > 0, "no batch was replayed request by request");
        CHECK(core.n_retried > 0, "no batch was retried after a device failure");
        CHECK(core.max_seen_batch > 1 && core.max_seen_batch <= 16, "max batch %ld", (long)core.max_seen_batch);
        long spread = 0;
        for (int64_t b : core.per_model_batches) spread += b > 0;
        CHECK(spread == 3, "only %ld of 3 models took batches", spread);
    }
    const long poisoned = [] { long n = 0; for (int k = 0; k < 64 * 30; ++k) n += k % 19 == 0; return n; }();
    CHECK(tot.wrong == 0, "%ld wrong results", tot.wrong.load());
    CHECK(tot.invalid == poisoned, "%ld INVALID results for %ld poison requests", tot.invalid.load(), poisoned);
    CHECK(tot.ok == 64 * 30 - poisoned, "%ld ok", tot.ok.load());
    CHECK(tot.device == 0 && tot.state == 0, "device %ld state %ld", tot.device.load(), tot.state.load());
}

void scenario_dead_model() {
    // the only model's every batch fails: the first batch is tried exactly twice (never once per request), then the model is
    // marked failed, what is queued reports DEVICE at once and later submits are refused without ever reaching a batch
    std::vector<StubHandle> hs = make_models(1);
    hs[0].m->dead = true;
    StubHandle* ptr[1] = {&hs[0]};
    ClientTotals tot;
    long batches, core_batches, model_failures;
    {
        Core core(ptr, 1, 32, 2000);
        std::vector<std::thread> th;
        for (int c = 0; c < 32; ++c)
            th.emplace_back([&, c] {
                unsigned rng = 99u + (unsigned)c;
                for (int i = 0; i < 4; ++i) one_request(core, rng, false, tot);
            });
        for (auto& t : th) t.join();
        std::lock_guard<std::mutex> lk(core.mu);
        core_batches = (long)core.n_batches;
        model_failures = (long)core.n_model_failures;
        CHECK(core.failed[0] == 1 && core.healthy_models() == 0, "the model is not marked failed");
    }
    batches = hs[0].m->batches.load();
    CHECK(tot.device == 128 && tot.ok == 0 && tot.wrong == 0, "device %ld ok %ld wrong %ld", tot.device.load(), tot.ok.load(), tot.wrong.load());
    CHECK(core_batches == 1 && batches == 2 && model_failures == 1, "%ld forwards for %ld batches, %ld model failures (want one batch, tried twice)", batches,
          core_batches, model_failures);
}

void scenario_one_faulted_model_of_three() {
    // Round 5 (the advisor's case): model 0 has a sticky fault -- every forward fails with DEVICE.  It must be fenced off after its
    // first batch (tried twice), that batch must be served by the healthy models, and NO request may fail; before, the dead
    // model counted as free at once after every failure and kept taking its share of the queue, failing each batch twice.
    std::vector<StubHandle> hs = make_models(3);
    hs[0].m->dead = true;
    StubHandle* ptr[3] = {&hs[0], &hs[1], &hs[2]};
    ClientTotals tot;
    {
        Core core(ptr, 3, 8, 300);
        std::vector<std::thread> th;
        for (int c = 0; c < 48; ++c)
            th.emplace_back([&, c] {
                unsigned rng = 4242u + (unsigned)c * 13u;
                for (int i = 0; i < 20; ++i) one_request(core, rng, false, tot);
            });
        for (auto& t : th) t.join();
        std::lock_guard<std::mutex> lk(core.mu);
        CHECK(core.failed[0] == 1 && core.failed[1] == 0 && core.failed[2] == 0, "health %d %d %d", core.failed[0], core.failed[1], core.failed[2]);
        CHECK(core.n_model_failures == 1 && core.n_requeued > 0 && core.n_retried == 1, "failures %ld requeued %ld retried %ld", (long)core.n_model_failures,
              (long)core.n_requeued, (long)core.n_retried);
        CHECK(core.per_model_batches[0] == 1, "the faulted model took %ld batches (want exactly one)", (long)core.per_model_batches[0]);
    }
    CHECK(hs[0].m->batches.load() == 2, "%ld forwards on the faulted model (want its one batch, tried twice)", hs[0].m->batches.load());
    CHECK(tot.ok == 48 * 20 && tot.device == 0 && tot.wrong == 0 && tot.state == 0, "ok %ld device %ld wrong %ld", tot.ok.load(), tot.device.load(),
          tot.wrong.load());
}

// Queue throughput with forwards that cost nothing: what the host side can dispatch per second in front of 8 models (the 8-GPU
// server of BASELINE configs[4] needs ~8 x 555 requests/s at the measured per-GPU rate).  Not a sanitizer scenario: built -O2.
int throughput() {
    std::vector<StubHandle> hs = make_models(8);
    StubHandle* ptr[8];
    for (int i = 0; i < 8; ++i) {
        hs[(size_t)i].m->sleep_us = 0;
        hs[(size_t)i].m->distribute_us = 0;
        ptr[i] = &hs[(size_t)i];
    }
    ClientTotals tot;
    const int clients = 64, per_client = 2000;
    const auto t0 = std::chrono::steady_clock::now();
    long batches;
    {
        Core core(ptr, 8, 64, 200);
        std::vector<std::thread> th;
        for (int c = 0; c < clients; ++c)
            th.emplace_back([&, c] {
                unsigned rng = 1u + (unsigned)c;
                for (int i = 0; i < per_client; ++i) one_request(core, rng, false, tot);
            });
        for (auto& t : th) t.join();
        std::lock_guard<std::mutex> lk(core.mu);
        batches = (long)core.n_batches;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("throughput: %.0f requests/s (%d clients x %d requests over 8 zero-cost models, %ld batches, %.3f s, %ld wrong)\n",
           clients * per_client / sec, clients, per_client, batches, sec, tot.wrong.load() + (long)(clients * per_client) - tot.ok.load());
    return (tot.ok == clients * per_client && tot.wrong == 0) ? 0 : 1;
}

void scenario_refused_at_submit() {
    std::vector<StubHandle> hs = make_models(2);
    hs[1].m->voices = 10;  // the smaller table decides
    StubHandle* ptr[2] = {&hs[0], &hs[1]};
    Core core(ptr, 2, 8, 100);
    std::vector<int64_t> ids = {0, 5, 6, 0};
    std::vector<float> style(KX_STYLE_DIM, 0.f);
    char err[256];
    void* out = nullptr;
    int64_t nb = 0, ns = 0;
    int32_t bad_v[2] = {3, 10};
    float w[2] = {0.5f, 0.5f};
    CHECK(core.submit_ex(ids.data(), 4, nullptr, bad_v, w, 2, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID && strstr(err, "voice id 10"), "voice id: %s", err);
    int32_t none_v[2] = {-1, -1};
    CHECK(core.submit_ex(ids.data(), 4, nullptr, none_v, w, 2, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID, "all parts skipped");
    CHECK(core.submit_ex(ids.data(), 4, nullptr, bad_v, w, 0, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID, "n_mix 0");
    CHECK(core.submit_ex(ids.data(), 4, nullptr, bad_v, w, 17, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID, "n_mix 17");
    CHECK(core.submit_ex(ids.data(), 1, nullptr, bad_v, nullptr, 1, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID, "one token with a voice");
    CHECK(core.submit_ex(ids.data(), 4, style.data(), bad_v, w, 2, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID, "row and voices");
    CHECK(core.submit_ex(ids.data(), 4, style.data(), nullptr, nullptr, 0, 0.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID, "speed 0");
    CHECK(core.submit_ex(ids.data(), 4, style.data(), nullptr, nullptr, 0, 1.f, 0, 3, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID, "format 3");
    std::vector<int64_t> bad_ids = {0, 178, 0};
    CHECK(core.submit_ex(bad_ids.data(), 3, style.data(), nullptr, nullptr, 0, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID && strstr(err, "0..177"), "token id: %s", err);
    hs[0].m->vocab = 100;  // the bound comes from the models' embedding tables (the smallest decides), not from a constant
    std::vector<int64_t> ids_100 = {0, 100, 0};
    CHECK(core.submit_ex(ids_100.data(), 3, style.data(), nullptr, nullptr, 0, 1.f, 0, 0, &out, &nb, &ns, err, sizeof err) == KX_ERR_INVALID && strstr(err, "0..99"), "token id vs a 100-row table: %s", err);
    hs[0].m->vocab = 178;
    int32_t ok_v[1] = {9};
    CHECK(core.submit_ex(ids.data(), 4, nullptr, ok_v, nullptr, 1, 1.f, 0, 2, &out, &nb, &ns, err, sizeof err) == KX_OK && nb == 4 * 10 * 2, "valid single voice: %s", err);
    free(out);
    std::lock_guard<std::mutex> lk(core.mu);
    CHECK(core.n_requests == 1, "refused requests reached a batch: %ld", (long)core.n_requests);
}

void scenario_destroy_while_queued() {
    // 64 clients block in submit with one slow model taking 4 at a time; the dispatcher is destroyed while most of them are
    // still queued: every queued request is served, nobody touches the object after it is gone
    for (int round = 0; round < 5; ++round) {
        std::vector<StubHandle> hs = make_models(1);
        hs[0].m->sleep_us = 1500;
        StubHandle* ptr[1] = {&hs[0]};
        ClientTotals tot;
        Core* core = new Core(ptr, 1, 4, 0);
        std::vector<std::thread> th;
        for (int c = 0; c < 64; ++c)
            th.emplace_back([&, c] {
                unsigned rng = 5u + (unsigned)c * 31u + (unsigned)round;
                one_request(*core, rng, false, tot);
            });
        for (;;) {  // until every client is inside submit() or already back from it (never counts a client too early)
            const long back = tot.ok + tot.invalid + tot.device + tot.state;
            std::unique_lock<std::mutex> lk(core->mu);
            if (core->inside + back >= 64) break;
            lk.unlock();
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        delete core;  // = kx_dispatcher_destroy
        for (auto& t : th) t.join();
        CHECK(tot.ok + tot.state == 64 && tot.wrong == 0, "round %d: ok %ld state %ld wrong %ld", round, tot.ok.load(), tot.state.load(), tot.wrong.load());
    }
    // clients that keep submitting until they are told the dispatcher is shutting down
    std::vector<StubHandle> hs = make_models(2);
    StubHandle* ptr[2] = {&hs[0], &hs[1]};
    ClientTotals tot;
    Core core(ptr, 2, 8, 200);
    std::vector<std::thread> th;
    for (int c = 0; c < 32; ++c)
        th.emplace_back([&, c] {
            unsigned rng = 77u + (unsigned)c;
            while (one_request(core, rng, false, tot) != KX_ERR_STATE) {
            }
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    core.shutdown();
    for (auto& t : th) t.join();
    CHECK(tot.state == 32 && tot.wrong == 0 && tot.ok > 0, "state %ld ok %ld wrong %ld", tot.state.load(), tot.ok.load(), tot.wrong.load());
}

void scenario_guard_and_last_error() {
    // kx::guarded / the thread-local last error: threads that fail differently beside threads that succeed on ONE handle
    StubHandle h;
    h.m.reset(new StubModel);
    StubHandle other;
    other.m.reset(new StubModel);
    std::atomic<int> bad{0};
    long shared_counter = 0;  // only ever touched under the model's mutex (a data race here is what TSan would report)
    std::vector<std::thread> th;
    for (int c = 0; c < 24; ++c)
        th.emplace_back([&, c] {
            for (int i = 0; i < 200; ++i) {
                const std::string mine = "failure of thread " + std::to_string(c) + " call " + std::to_string(i);
                int rc;
                if (c % 3 == 0) {
                    rc = kx::guarded(&h, [&](StubModel&) { ++shared_counter; });
                    if (rc != 0 || *kx::last_error_of(&h)) ++bad;
                } else if (c % 3 == 1) {
                    rc = kx::guarded(&h, [&](StubModel&) { ++shared_counter; throw kx::Error(1, mine); });
                    if (rc != 1 || mine != kx::last_error_of(&h) || *kx::last_error_of(&other)) ++bad;
                } else {
                    rc = kx::guarded(&h, [&](StubModel&) { ++shared_counter; throw std::runtime_error(mine); });
                    if (rc != 3 || mine != kx::last_error_of(&h)) ++bad;
                    rc = kx::guarded(&h, [&](StubModel&) { ++shared_counter; });  // own success clears own message only
                    if (rc != 0 || *kx::last_error_of(&h)) ++bad;
                }
            }
        });
    for (auto& t : th) t.join();
    CHECK(bad == 0, "%d wrong last-error observations", bad.load());
    CHECK(shared_counter == 8 * 200 + 8 * 200 + 8 * 400, "counter %ld", shared_counter);
    StubHandle* null_h = nullptr;
    CHECK(kx::guarded(null_h, [&](StubModel&) {}) == 1, "null handle");
    char buf[8];
    CHECK(kx::guarded_free(buf, sizeof buf, [] { throw kx::Error(2, "a long message that must be cut"); }) == 2 && strlen(buf) == 7, "guarded_free: '%s'", buf);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "throughput") == 0) return throughput();
    scenario_guard_and_last_error();
    scenario_refused_at_submit();
    scenario_mixed_load();
    scenario_dead_model();
    scenario_one_faulted_model_of_three();
    scenario_destroy_while_queued();
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("host concurrency scenarios passed\n");
    return 0;
}
