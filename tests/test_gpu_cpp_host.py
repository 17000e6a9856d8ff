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
