"""GPU: a blob written by the ONNX importer drives the HIP forward.

The exporter-style `model.onnx` files of tests/test_importer_onnx.py (fp32 with weight norm kept / folded / anonymous
weights, int8 MatMulInteger + DequantizeLinear, 4-bit MatMulNBits: the variants of kokorox/src/utils/hf_cache.rs:135-144)
are imported to .kxw, loaded through kx_create, and the waveform is compared with the CPU oracle loaded from the SAME
imported blob under the usual protocol (F0 / N curves and source STFT pinned): the quantised
variants reproduce their de-quantised weights, not ONNX Runtime's integer arithmetic."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dresser():
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_importer_onnx.py")
    spec = importlib.util.spec_from_file_location("_importer_cases", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _hip_vs_oracle(m, o, ids, style, seed):
    from kokorox_amd import hip_koko as hk
    out = m.infer([list(ids)], [list(style)], 1.0, seed=seed, flags=hk.KX_FLAG_TAPS)
    taps = {}
    _, dur = o.forward(ids, style, 1.0, seed=seed, utt=0, taps=taps)
    assert out.shape[0] == 600 * int(dur.sum()), "durations differ from the oracle's"
    for name in ("bert.out", "d_en", "dur.lstm", "text_enc.out", "dec.decode.3"):
        ref = taps[name].numpy()
        assert np.abs(m.tap(name, 0) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name
    f0, n_c, har = m.tap("pred.F0", 0), m.tap("pred.N", 0)[0], m.tap("gen.har", 0)
    ref_f0 = taps["pred.F0"].numpy()
    assert np.abs(f0 - ref_f0).max() <= 5e-5 * np.abs(ref_f0).max()
    audio, _ = o.forward(ids, style, 1.0, seed=seed, utt=0, f0_override=f0[0], n_override=n_c, har_override=har)
    return out, float(np.abs(out - audio.numpy()).max())


@pytest.mark.parametrize("style", ["fp32", "int8", "q4"])
def test_imported_onnx_blob_runs_and_matches_the_oracle(hip_model, blob_path, tmp_path, style):
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import importer as I
    from kokorox_amd import onnx_lite as OX
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    C = _dresser()
    synth = W.read_blob(blob_path)
    nodes, inits = C._dress_as_export({k: np.asarray(v) for k, v in synth.items()}, style)
    src = str(tmp_path / f"model_{style}.onnx")
    with open(src, "wb") as f:
        f.write(OX.model_bytes(nodes, inits))
    dst = str(tmp_path / f"model_{style}.kxw")
    I.import_checkpoint(src, dst)
    ids = R.synthetic_inputs(1, 16, seed=77)[0]
    style_row = W.synthetic_voices(1)[0, 16, 0]
    m = hk.HipKoko.new(dst)
    try:
        o = R.KokoroOracle(dst)
        out, err = _hip_vs_oracle(m, o, ids, style_row, seed=6)
        assert err < 1e-4, f"{style}: waveform differs from the oracle on the same imported blob: {err}"
        ref = hip_model.infer([list(ids)], [list(style_row)], 1.0, seed=6)
        if style == "fp32":
            # the fp32 export carries the same weights (weight-norm folds re-rounded at 2e-6): same durations, and the
            # waveform equals the original blob's up to the F0-phase sensitivity documented in DESIGN.md section 4
            assert out.shape == ref.shape
        else:
            # quantised weights are different weights: the result must differ from the fp32 model's (the de-quantised
            # values really reached the kernels) while matching the oracle that holds the same de-quantised blob
            assert out.shape != ref.shape or np.abs(out - ref).max() > 1e-3
    finally:
        m.close()


# ---- kx_create / kx_create_replicas handed the .onnx itself (what OrtKoko::new receives, koko.rs:570-573) --------------
def _write_onnx(synth, path, style):
    from kokorox_amd import onnx_lite as OX
    C = _dresser()
    nodes, inits = C._dress_as_export({k: np.asarray(v) for k, v in synth.items()}, style)
    with open(path, "wb") as f:
        f.write(OX.model_bytes(nodes, inits))


def _fnv1a(a: np.ndarray) -> int:
    h = 1469598103934665603
    for byte in a.astype("<f4").tobytes():
        h ^= byte
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("style", ["fp32", "int8", "q4"])
def test_kx_create_opens_the_onnx_file_itself(blob_path, golden, tmp_path, style):
    """`HipKoko::new(model_path)` with the path the reference passes: the library's own ONNX reader (csrc/onnx_import.cpp)
    must give the waveform of the Python-imported .kxw route bit for bit, from ctypes and from compiled C++."""
    import subprocess
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import importer as I
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    src = str(tmp_path / f"model_{style}.onnx")
    _write_onnx(W.read_blob(blob_path), src, style)
    kxw = str(tmp_path / f"model_{style}.python.kxw")
    I.import_checkpoint(src, kxw)
    ids = R.synthetic_inputs(1, 16, seed=77)[0]
    style_row = W.synthetic_voices(1)[0, 16, 0]
    m_py = hk.HipKoko.new(kxw)
    try:
        ref = m_py.infer([list(ids)], [list(style_row)], 1.0, seed=6)
        g = golden["hello_world"]
        ref_g = m_py.infer([list(g["ids"])], [list(g["style"])], 1.0, seed=2)
    finally:
        m_py.close()
    m = hk.HipKoko.new(src)  # the .onnx path, as koko.rs:570-573 passes it
    try:
        out = m.infer([list(ids)], [list(style_row)], 1.0, seed=6)
        info = m.info()
    finally:
        m.close()
    np.testing.assert_array_equal(out, ref)
    # kx_model_info names what was loaded, and says that the quantised variants do NOT run the reference's arithmetic (ORT
    # quantises the activations at run time; this library de-quantises the weights and runs f32-class)
    want = {"fp32": (1, True), "int8": (3, False), "q4": (4, False)}[style]
    assert (info["variant"], info["reference_arithmetic"]) == want, info
    assert info["n_vocab"] == 178 and info["cu_partitions"] == 1 and info["cus"] == 256, info
    assert not os.path.exists(src + ".kxw"), "no cache file may appear unasked"
    # compiled C++ host on the same .onnx
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "kokorox_amd", "lib")
    exe = str(tmp_path / "hipkoko_demo")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "hipkoko_demo.cpp"), "-L", lib_dir, "-lkokorox_hip",
                    f"-Wl,-rpath,{lib_dir}", "-o", exe], check=True)
    srow = str(tmp_path / "style.f32")
    g["style"].astype("<f4").tofile(srow)
    r = subprocess.run([exe, src, srow], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"samples={ref_g.shape[0]} fnv1a={_fnv1a(ref_g):016x}" in r.stdout


def test_onnx_cache_replicas_and_error_paths(blob_path, tmp_path, monkeypatch):
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    src = str(tmp_path / "model.onnx")
    _write_onnx(W.read_blob(blob_path), src, "fp32")
    ids = R.synthetic_inputs(1, 12, seed=5)[0]
    style_row = W.synthetic_voices(1)[0, 12, 0]
    # replicas from the .onnx: one read + one conversion, two models
    ms = hk.HipKoko.replicas(src, [0, 0])
    try:
        a = ms[0].infer([list(ids)], [list(style_row)], 1.0, seed=1)
        b = ms[1].infer([list(ids)], [list(style_row)], 1.0, seed=1)
    finally:
        for m in ms:
            m.close()
    np.testing.assert_array_equal(a, b)
    # opt-in cache (KOKOROX_KXW_CACHE=1): written beside the .onnx with a stamp naming the source's size, mtime (ns) and the
    # importer's version; consulted only under the same switch and only while the stamp matches exactly
    monkeypatch.setenv("KOKOROX_KXW_CACHE", "1")
    m = hk.HipKoko.new(src)
    m.close()
    cache = src + ".kxw"
    assert os.path.exists(cache) and os.path.exists(cache + ".src")
    py = str(tmp_path / "py.kxw")
    from kokorox_amd import importer as I
    I.import_onnx(src, py)
    assert open(cache, "rb").read() == open(py, "rb").read()
    # corrupt the .onnx but keep its size and mtime: the stamp still matches, the cache is what loads
    st = os.stat(src)
    onnx_bytes = bytearray(open(src, "rb").read())
    onnx_bytes[0:64] = b"\xff" * 64  # (the ModelProto's first fields: no longer parseable)
    with open(src, "wb") as f:
        f.write(onnx_bytes)
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns))
    m = hk.HipKoko.new(src)
    try:
        np.testing.assert_array_equal(m.infer([list(ids)], [list(style_row)], 1.0, seed=1), a)
    finally:
        m.close()
    # without the switch a cache file is never trusted, however new: the (corrupt) .onnx is read
    monkeypatch.delenv("KOKOROX_KXW_CACHE")
    with pytest.raises(RuntimeError, match="not a readable ONNX model|ONNX"):
        hk.HipKoko.new(src)
    # with the switch, a source whose mtime moved -- EARLIER, as cp -p / mv / a re-pointed blob symlink leave it -- no longer
    # matches the stamp: the cache is ignored (a "cache not older than the source" test would have loaded it)
    monkeypatch.setenv("KOKOROX_KXW_CACHE", "1")
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns - 10_000_000_000))
    with pytest.raises(RuntimeError, match="not a readable ONNX model|ONNX"):
        hk.HipKoko.new(src)
    monkeypatch.delenv("KOKOROX_KXW_CACHE")
    with pytest.raises(RuntimeError, match="not a readable ONNX model"):
        hk.HipKoko.replicas(src, [0, 0])


def test_four_replicas_on_one_gpu_build_fail_and_destroy(blob_path, monkeypatch):
    """BASELINE configs[3] / [4] without the 8-GPU node: kx_create_replicas with the one device id named four times --
    four CU-partitioned models from ONE file read, built one host thread each; they give the bits of a whole-device model; they
    can be destroyed in any order; and when the build of replica 2 fails (injected) nothing is leaked and every handle is NULL.
    Distinct-device operation (peer copies over xGMI) stays unverified until an N-GPU run exists."""
    import torch
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    ids = R.synthetic_inputs(1, 20, seed=41)[0]
    style_row = W.synthetic_voices(1)[0, 20, 0]
    whole = hk.HipKoko.new(blob_path)
    try:
        ref = whole.infer([list(ids)], [list(style_row)], 1.0, seed=3)
        assert whole.info()["cu_partitions"] == 1
    finally:
        whole.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for order in ((0, 1, 2, 3), (3, 1, 0, 2)):
        ms = hk.HipKoko.replicas(blob_path, [0, 0, 0, 0])
        t_read, t_resident, t_built = hk.HipKoko.replicas_times()
        assert 0 < t_read <= t_resident <= t_built, (t_read, t_resident, t_built)
        try:
            infos = [m.info() for m in ms]
            assert [(i["cu_partition"], i["cu_partitions"], i["cus"]) for i in infos] == [(k, 4, 64) for k in range(4)], infos
            for m in ms:
                np.testing.assert_array_equal(m.infer([list(ids)], [list(style_row)], 1.0, seed=3), ref)
        finally:
            for k in order:
                ms[k].close()
    # a failed build in the middle: RuntimeError, every handle gone, the device memory back
    monkeypatch.setenv("KX_TEST_HOOKS", "1")
    monkeypatch.setenv("KX_TEST_FAIL_REPLICA", "2")
    with pytest.raises(RuntimeError, match="replica 2"):
        hk.HipKoko.replicas(blob_path, [0, 0, 0, 0])
    monkeypatch.delenv("KX_TEST_FAIL_REPLICA")
    monkeypatch.delenv("KX_TEST_HOOKS")
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) >> 20} MiB of device memory did not come back"
