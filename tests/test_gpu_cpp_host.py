"""GPU: the C ABI driven from compiled C++ (include/kokorox_hip.hpp) agrees bit-for-bit with ctypes."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv1a(a: np.ndarray) -> int:
    h = 1469598103934665603
    for byte in a.astype("<f4").tobytes():
        h ^= byte
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_host_matches_ctypes(hip_model, blob_path, golden, tmp_path):
    lib_dir = os.path.join(ROOT, "kokorox_amd", "lib")
    exe = str(tmp_path / "hipkoko_demo")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "hipkoko_demo.cpp"), "-L", lib_dir, "-lkokorox_hip",
                    f"-Wl,-rpath,{lib_dir}", "-o", exe], check=True)
    g = golden["hello_world"]
    style = str(tmp_path / "style.f32")
    g["style"].astype("<f4").tofile(style)
    r = subprocess.run([exe, blob_path, style], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ref = hip_model.infer([list(g["ids"])], [list(g["style"])], 1.0, seed=2)
    assert f"samples={ref.shape[0]} fnv1a={_fnv1a(ref):016x}" in r.stdout
    assert "empty-input=error" in r.stdout


def test_cpp_replicas_and_dispatcher_over_two_models(hip_model, blob_path, golden, tmp_path):
    """kx_create_replicas (one file read, n models) + a dispatcher over more than one model, from compiled C++.
    On the 1-GPU box both models live on device 0; every request must equal the same request run alone."""
    lib_dir = os.path.join(ROOT, "kokorox_amd", "lib")
    exe = str(tmp_path / "replicas_demo")
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "replicas_demo.cpp"), "-L", lib_dir,
                    "-lkokorox_hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib",
                    "-lpthread", "-o", exe], check=True)
    g = golden["hello_world"]
    style = str(tmp_path / "style.f32")
    g["style"].astype("<f4").tofile(style)
    try:
        r = subprocess.run([exe, blob_path, style, "2"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, KX_TRACE_DTOR="1"))
    except subprocess.TimeoutExpired as e:  # (say where it stalled: the demo and the model destructor report their stages on stderr)
        err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        raise AssertionError("replicas demo stalled; stderr so far:\n" + err[-3000:] + "\nstdout so far:\n" + out[-1500:])
    assert r.returncode == 0, r.stderr
    assert "models=2 requests=8" in r.stdout
    row = [0, 50, 83, 54, 156, 57, 135, 3, 16, 65, 156, 87, 158, 54, 46, 5, 0]
    hip_model.set_utterance_base(0)
    for i in range(8):
        ids = row[:5 + i] + [0]
        ref = hip_model.infer([ids], [list(g["style"])], 1.0, seed=100 + i)
        assert f"req{i} rc=0 samples={ref.shape[0]} fnv1a={_fnv1a(ref):016x}" in r.stdout, r.stdout
