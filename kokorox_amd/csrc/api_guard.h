// The exception fence of the C ABI and the per-thread last-error text (HIP-free: api.hip uses it on kx_model, the CPU
// suite's sanitizer build on a stub handle, tests/cpp/host_sanitize.cpp).
//
// What the reference hand-asserts with `unsafe impl Send / Sync for OrtKoko` over a `Mutex<Session>`
// (kokorox/src/onn/ort_koko.rs:14,17-18,78) is enforced here: calls on one model are serialised by the model's mutex,
// nothing throws across the ABI, and the message of a failed call belongs to the THREAD that made the call.
#pragma once
#include <cstring>
#include <mutex>
#include <string>

#include "kx_error.h"

namespace kx {

inline void set_err(char* err, size_t n, const std::string& msg) {
    if (err && n) {
        strncpy(err, msg.c_str(), n - 1);
        err[n - 1] = 0;
    }
}

// Several threads may share one model (the Rust handle is Send + Sync): the text is kept per thread, set by that thread's
// own failure and cleared by its own next success on the same model, so another thread's calls can neither blank it nor
// replace it.
struct LastError {
    std::string text;
    const void* model = nullptr;
};
inline LastError& tls_last_error() {
    static thread_local LastError e;
    return e;
}
inline void note_error(const void* h, const std::string& msg) {
    LastError& e = tls_last_error();
    e.text = msg;
    e.model = h;
}
inline const char* last_error_of(const void* h) {
    LastError& e = tls_last_error();
    return e.model == h ? e.text.c_str() : "";
}

// status codes of include/kokorox_hip.h (KX_OK, KX_ERR_INVALID, KX_ERR_DEVICE) by value: this header has no C ABI include
template <class Handle, class F>
int guarded(Handle* h, F&& f) {
    if (!h || !h->m) return 1;
    std::lock_guard<std::mutex> lk(h->m->mu);
    try {
        f(*h->m);
        LastError& e = tls_last_error();
        if (e.model == h) {
            e.text.clear();
            e.model = nullptr;
        }
        return 0;
    } catch (const Error& e) {
        note_error(h, e.what());
        return e.code;
    } catch (const std::exception& e) {
        note_error(h, e.what());
        return 3;
    } catch (...) {
        note_error(h, "unknown failure");
        return 3;
    }
}

template <class F>
int guarded_free(char* err, size_t n, F&& f) {
    try {
        f();
        return 0;
    } catch (const Error& e) {
        set_err(err, n, e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_err(err, n, e.what());
        return 3;
    } catch (...) {
        set_err(err, n, "unknown failure");
        return 3;
    }
}

}  // namespace kx
