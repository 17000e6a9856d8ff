"""CPU: the library's host-side concurrency under ThreadSanitizer and AddressSanitizer + UBSan (g++; no GPU sanitizer
exists on this pool).

The reference hand-asserts thread safety (`unsafe impl Send / Sync for OrtKoko` over a `Mutex<Session>`,
/root/reference/kokorox/src/onn/ort_koko.rs:14,17-18,78).  This library adds real concurrency on the host — dispatcher
workers and condition variables (kokorox_amd/csrc/dispatcher_core.h), the exception fence and the thread-local last error
of the C ABI (api_guard.h) — and both headers are HIP-free so that tests/cpp/host_sanitize.cpp can drive them against a stub
model: 64 client threads, mixed voices and output formats, batches failing as a whole (INVALID -> replayed one by one,
DEVICE -> one retry), a model that stays broken, requests refused at submit, destroy-while-queued.
Not covered here: the pooled page-locked buffers (HIP allocations, model.hip) — they are exercised by the GPU suite only.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,flags,env", [
    ("thread", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"}),
    ("address", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"],
     {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1"}),
])
def test_dispatcher_and_api_guard_under_sanitizers(tmp_path, name, flags, env):
    exe = str(tmp_path / f"host_{name}")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", *flags, "-I", os.path.join(ROOT, "kokorox_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), "-o", exe, "-lpthread"], check=True)
    for _ in range(3):  # (thread interleavings differ from run to run)
        r = subprocess.run([exe], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "scenarios passed" in r.stdout, r.stdout[-1000:] + r.stderr[-6000:]
        assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr


def test_dispatcher_queue_throughput_with_zero_cost_models(tmp_path):
    """What the dispatcher's host side can move per second when the forwards cost nothing: 64 client threads in front of 8 stub
    models (optimised build, no sanitizer).  The 8-GPU server of BASELINE configs[4] needs 8 x ~555 requests/s at the per-GPU
    rate measured on one MI355X: the queue must not be what caps it.  The number is printed for DESIGN.md."""
    exe = str(tmp_path / "host_throughput")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "kokorox_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), "-o", exe, "-lpthread"], check=True)
    r = subprocess.run([exe, "throughput"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    m = re.search(r"throughput: (\d+) requests/s", r.stdout)
    assert m, r.stdout
    print(r.stdout.strip())
    assert int(m.group(1)) >= 5000, r.stdout
