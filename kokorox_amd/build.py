"""Build libkokorox_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libkokorox_hip.so")
# the stand-alone kernel hooks of tests/ (include/kokorox_hip_test.h): a library of its own, linked against the one above
TEST_LIB_PATH = os.path.join(LIB_DIR, "libkokorox_hip_test.so")
TEST_SOURCES = ["test_hooks.hip"]
SOURCES = ["conv_mfma.hip", "conv_f16x3.hip", "conv_f16x3_da.hip", "conv_f16x3_da_p1.hip", "conv_f16x3_da_w2.hip", "conv_f16x3_da_s16.hip", "conv_f16x3_da_f8.hip", "conv_f16x3_da_pre.hip", "conv_f16x3_pre.hip", "conv_f16x3_dapn.hip", "conv_f16x3_dag.hip", "kernels_misc.hip", "model.hip", "api.hip", "dispatcher.hip", "onnx_import.cpp"]
HEADERS = ["kx_common.h", "kx_error.h", "conv_epilogue.h", "conv_f16x3_common.h"]
_PUB = os.path.join("..", "..", "include", "kokorox_hip.h")
_PUB_TEST = os.path.join("..", "..", "include", "kokorox_hip_test.h")
# sources that include another source (the direct-A instantiation units) or a header of their own: rebuilt when that one changes
EXTRA_DEPS = {"model.hip": ["onnx_import.h", "model.h", _PUB], "api.hip": ["model.h", "kx_handle.h", "api_guard.h", _PUB],
              "dispatcher.hip": ["model.h", "kx_handle.h", "dispatcher_core.h", _PUB],
              "test_hooks.hip": ["model.h", "kx_handle.h", "api_guard.h", _PUB, _PUB_TEST], "onnx_import.cpp": ["onnx_import.h"], "conv_f16x3_da_p1.hip": ["conv_f16x3_da.hip"], "conv_f16x3_da_w2.hip": ["conv_f16x3_da.hip"], "conv_f16x3_da_s16.hip": ["conv_f16x3_da.hip"], "conv_f16x3_da_f8.hip": ["conv_f16x3_da.hip"], "conv_f16x3_da_pre.hip": ["conv_f16x3_da.hip"]}
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


LAST_BUILD = {"compiled": [], "reused": [], "linked": False}  # what the last build_library() call did (build() reports it)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit and link the shared library; returns its path."""
    LAST_BUILD["compiled"], LAST_BUILD["reused"], LAST_BUILD["linked"] = [], [], False
    os.makedirs(LIB_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, test_objs, jobs = [], [], []
    for src in SOURCES + TEST_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        (test_objs if src in TEST_SOURCES else objs).append(o)
        if force or not _newer(o, [s] + hdrs + [os.path.join(CSRC, d) for d in EXTRA_DEPS.get(src, [])]):
            # plain C++ sources (the ONNX reader) are host code only: no offload flag
            flags = [f for f in FLAGS if not f.startswith("--offload-arch")] if src.endswith(".cpp") else FLAGS
            jobs.append([_hipcc(), *flags, "-c", s, "-o", o])
            LAST_BUILD["compiled"].append(src)
        else:
            LAST_BUILD["reused"].append(src)
    if jobs:
        # the translation units are independent: compile them side by side (conv_f16x3.hip alone is half the time)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)

        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or not _newer(LIB_PATH, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lpthread", "-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        LAST_BUILD["linked"] = True
    if force or not _newer(TEST_LIB_PATH, test_objs + [LIB_PATH]):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *test_objs, "-L" + LIB_DIR, "-lkokorox_hip", "-Wl,-rpath,$ORIGIN",
               "-lpthread", "-o", TEST_LIB_PATH]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
