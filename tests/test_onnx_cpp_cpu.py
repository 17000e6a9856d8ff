"""CPU: the in-library ONNX reader (kokorox_amd/csrc/onnx_import.cpp, behind kx_create / kx_import_onnx).

`OrtKoko::new(model_path)` is handed the Hugging Face `.onnx` (/root/reference/kokorox/src/tts/koko.rs:570-573 ->
onn/ort_base.rs:27-33 `commit_from_file`; path built at utils/hf_cache.rs:128-158, variants hf_cache.rs:135-144), so
the library has to open that file itself.  Checked here without a GPU (kx_import_onnx is host only):

  * on the exporter-style files of test_importer_onnx.py (fp32 with weight norm kept / folded / anonymous weights, fp16,
    int8 MatMulInteger + DequantizeLinear, 4-bit MatMulNBits) the C++ reader writes the SAME BYTES as the Python importer;
  * unplaced initialisers are listed in the error; truncated files, external-data tensors and non-ONNX bytes fail with
    KX_ERR_IO (never a crash);
  * the walker is fuzzed under AddressSanitizer + UBSan (g++): random truncations, bit flips and bad varints never read
    out of bounds (tests/cpp/onnx_fuzz.cpp).

No exporter-written `.onnx` has ever been read by either importer (none exists offline): these files are serialised by
the repo's own writer in the exporter's conventions as recalled, see INTEGRATION.md.
"""
import importlib.util
import os
import subprocess

import numpy as np
import pytest

from kokorox_amd import hip_koko as hk
from kokorox_amd import importer as I
from kokorox_amd import onnx_lite as OX
from kokorox_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dresser():
    p = os.path.join(ROOT, "tests", "test_importer_onnx.py")
    spec = importlib.util.spec_from_file_location("_importer_cases", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def synth():
    return {k: np.asarray(v) for k, v in W.read_blob(W.ensure_synthetic_blob()).items()}


@pytest.fixture(scope="module")
def files(synth, tmp_path_factory):
    """style -> path of an exporter-style model_<style>.onnx"""
    d = tmp_path_factory.mktemp("onnx_cpp")
    C = _dresser()
    out = {}
    for style in ("fp32", "fp16", "int8", "q4"):
        nodes, inits = C._dress_as_export(synth, style)
        p = str(d / f"model_{style}.onnx")
        with open(p, "wb") as f:
            f.write(OX.model_bytes(nodes, inits))
        out[style] = p
    out["dir"] = str(d)
    return out


@pytest.mark.parametrize("style", ["fp32", "fp16", "int8", "q4"])
def test_cpp_reader_writes_the_python_importers_bytes(files, style):
    src = files[style]
    py = src + ".py.kxw"
    cpp = src + ".cpp.kxw"
    I.import_onnx(src, py)
    hk.import_onnx(src, cpp)
    a = np.fromfile(py, dtype=np.uint8)
    b = np.fromfile(cpp, dtype=np.uint8)
    assert a.shape == b.shape
    if not np.array_equal(a, b):  # name the first tensor that differs
        ta, tb = W.read_blob(py), W.read_blob(cpp)
        assert list(ta) == list(tb)
        for k in ta:
            np.testing.assert_array_equal(np.asarray(ta[k]), np.asarray(tb[k]), err_msg=k)
        raise AssertionError("blobs differ outside the tensor data")
    os.remove(py)
    os.remove(cpp)


def test_cpp_spec_is_the_python_spec(files):
    """the table compiled into the library = kokorox_amd.weights.tensor_spec(): names, order, shapes (read back from a
    converted blob)."""
    dst = files["fp32"] + ".spec.kxw"
    hk.import_onnx(files["fp32"], dst)
    got = W.read_blob(dst)
    spec = W.tensor_spec()
    assert list(got) == list(spec)
    for k, (shape, _) in spec.items():
        assert tuple(got[k].shape) == tuple(shape), k
    os.remove(dst)


def test_missing_tensors_and_unplaced_initialisers_are_named(synth, files):
    C = _dresser()
    nodes, inits = C._dress_as_export(synth, "fp32")
    keep = [n for n in nodes if b"/predictor/lstm/LSTM" not in n]
    src = os.path.join(files["dir"], "broken.onnx")
    with open(src, "wb") as f:
        f.write(OX.model_bytes(keep, inits))
    with pytest.raises(hk.KokoroxHipError) as ei:
        hk.import_onnx(src, src + ".kxw")
    assert ei.value.code == hk.KX_ERR_IO
    msg = str(ei.value)
    assert "predictor.lstm" in msg and "could not place" in msg and "onnx::LSTM" in msg
    assert not os.path.exists(src + ".kxw")


def test_bad_files_fail_with_io_status(files, tmp_path):
    data = open(files["fp32"], "rb").read()
    cases = {
        "truncated_half": data[: len(data) // 2],
        "truncated_tail": data[:-7],
        "not_onnx": b"NOTABLOB" + b"\0" * 100,
        "no_graph": b"\x08\x08",
        "empty": b"",
        "varint_runaway": b"\xff" * 64,
    }
    # one tensor declared with external data (data_location = EXTERNAL)
    ext = OX._vi(1, 4) + OX._vi(2, OX.FLOAT) + OX._ld(8, b"w") + OX._vi(14, 1)
    cases["external_data"] = OX.model_bytes([], [ext])
    for name, blob in cases.items():
        p = tmp_path / f"{name}.onnx"
        p.write_bytes(blob)
        with pytest.raises(hk.KokoroxHipError) as ei:
            hk.import_onnx(str(p), str(tmp_path / "out.kxw"))
        assert ei.value.code == hk.KX_ERR_IO, name
        if name == "external_data":
            assert "external data" in str(ei.value)
    with pytest.raises(hk.KokoroxHipError, match="cannot open") as ei:
        hk.import_onnx(str(tmp_path / "absent.onnx"), str(tmp_path / "out.kxw"))
    assert ei.value.code == hk.KX_ERR_IO


def test_walker_fuzz_under_address_and_ub_sanitizers(files, tmp_path):
    """g++ -fsanitize=address,undefined build of the reader alone; a sanitizer report aborts the binary (non-zero exit)."""
    exe = str(tmp_path / "onnx_fuzz")
    csrc = os.path.join(ROOT, "kokorox_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", csrc, os.path.join(ROOT, "tests", "cpp", "onnx_fuzz.cpp"), os.path.join(csrc, "onnx_import.cpp"),
                    "-o", exe], check=True)
    # Small files keep thousands of mutations fast.  The walker's paths depend on node kinds and encodings, not on tensor
    # sizes: every big tensor is cut down to a corner of itself (same name and rank, so its node is still found and its
    # placement rules still run; the import then ends in the shape check), one LSTM keeps its recurrent matrix whole
    # because the LSTM branch checks it against hidden_size.  The full q4 file adds a few mutations that reach the blob
    # writer.
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    C = _dresser()
    synth = {k: np.asarray(v) for k, v in W.read_blob(W.ensure_synthetic_blob()).items()}
    cut = {}
    for k, v in synth.items():
        if (".lstm" in k or k.startswith("predictor.shared.")) and not k.startswith("text_encoder.lstm."):
            continue
        if v.size <= 4096:
            cut[k] = v
        elif k.startswith("text_encoder.lstm.weight_ih"):
            cut[k] = np.ascontiguousarray(v[:, :4])
        elif k.startswith("text_encoder.lstm.weight_hh"):
            cut[k] = v
        else:
            cut[k] = np.ascontiguousarray(v[tuple(slice(0, min(d, 8 if i < 2 else d)) for i, d in enumerate(v.shape))])
    procs = []
    for style, n_mut in (("fp32", 1500), ("int8", 1000), ("q4", 800)):  # side by side: three of the 8 cores
        nodes, inits = C._dress_as_export(cut, style, pooler=16)
        small = str(tmp_path / f"small_{style}.onnx")
        with open(small, "wb") as f:
            f.write(OX.model_bytes(nodes, inits))
        assert os.path.getsize(small) < 4 << 20
        procs.append(subprocess.Popen([exe, small, str(n_mut), "1"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    procs.append(subprocess.Popen([exe, files["q4"], "6", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                  text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
        assert "mutations" in so
    assert " 0 converted" not in outs[-1][0]  # the unmutated full file converts under the sanitizers too


def test_less_common_encodings_agree_between_the_two_readers(synth, tmp_path):
    """Branches the exporter-style files above do not reach, on a model whose other tensors are plain named initialisers:
    Gemm (transB = 1, bias as third input), a weight that reaches its MatMul through Cast and through Transpose, a weight in
    a Constant node, ConvInteger with the `_quantized / _scale / _zero_point` naming, DequantizeLinear with block_size,
    bfloat16 and double initialisers, float16 carried in int32_data, a DataParallel-style 'module.' prefix, a wrapped
    model ('kmodel.' prefix on an initialiser, '/kmodel/...' and '/model/kmodel/...' on node names).  The C++ reader and the Python importer must write the same bytes, and the values must be the
    expected ones."""
    t = {k: v.copy() for k, v in synth.items()}
    nodes, inits = [], []
    special = {}

    def named(name, a, **kw):
        inits.append(OX.tensor_bytes(name, a, **kw))

    # 1. Gemm with transB = 1: weight kept as [out, in], bias third input
    k = "bert_encoder.weight"
    special[k] = t[k]
    named("onnx::Gemm_1", t[k])
    named("onnx::Gemm_2", t["bert_encoder.bias"])
    nodes.append(OX.node_bytes("Gemm", ["x", "onnx::Gemm_1", "onnx::Gemm_2"], ["g"], name="/bert_encoder/Gemm", attrs={"transB": 1}))
    # 2. fp16 initialiser -> Cast -> MatMul (weight stored transposed), named bias on the Add
    k = "predictor.duration_proj.linear_layer.weight"
    special[k] = t[k].astype(np.float16).astype(np.float32)
    named("onnx::Cast_3", np.ascontiguousarray(t[k].T).astype(np.float16))
    nodes.append(OX.node_bytes("Cast", ["onnx::Cast_3"], ["c3"], name="/predictor/duration_proj/linear_layer/Cast", attrs={"to": 1}))
    nodes.append(OX.node_bytes("MatMul", ["x", "c3"], ["mm3"], name="/predictor/duration_proj/linear_layer/MatMul"))
    named("predictor.duration_proj.linear_layer.bias", t["predictor.duration_proj.linear_layer.bias"])
    nodes.append(OX.node_bytes("Add", ["predictor.duration_proj.linear_layer.bias", "mm3"], ["o3"], name="/predictor/duration_proj/linear_layer/Add"))
    # 3. initialiser stored [out, in] -> Transpose -> MatMul; the node names carry two extra leading components
    k = "bert.encoder.embedding_hidden_mapping_in.weight"
    special[k] = t[k]
    named("onnx::Transpose_4", t[k])
    nodes.append(OX.node_bytes("Transpose", ["onnx::Transpose_4"], ["t4"], name="/model/kmodel/bert/encoder/embedding_hidden_mapping_in/Transpose",
                               attrs={"perm": [1, 0]}))
    nodes.append(OX.node_bytes("MatMul", ["x", "t4"], ["mm4"], name="/model/kmodel/bert/encoder/embedding_hidden_mapping_in/MatMul"))
    named("bert.encoder.embedding_hidden_mapping_in.bias", t["bert.encoder.embedding_hidden_mapping_in.bias"])
    nodes.append(OX.node_bytes("Add", ["mm4", "bert.encoder.embedding_hidden_mapping_in.bias"], ["o4"],
                               name="/bert/encoder/embedding_hidden_mapping_in/Add"))
    # 4. a conv weight that lives in a Constant node (attribute tensor), bias anonymous: both through the node name
    k = "decoder.asr_res.0.weight"
    special[k] = t[k]
    cattr = OX._ld(1, b"value") + OX._ld(5, OX.tensor_bytes("", t[k])) + OX._vi(20, 4)
    nodes.append(OX._ld(2, b"const5") + OX._ld(3, b"/decoder/asr_res.0/Constant") + OX._ld(4, b"Constant") + OX._ld(5, cattr))
    named("onnx::Conv_6", t["decoder.asr_res.0.bias"])
    nodes.append(OX.node_bytes("Conv", ["x", "const5", "onnx::Conv_6"], ["o5"], name="/kmodel/decoder/asr_res.0/Conv"))
    # 5. ConvInteger, per-tensor scale and zero point
    k = "decoder.generator.noise_convs.1.weight"
    sc = np.float32(np.abs(t[k]).max() / 127.0)
    q = np.clip(np.round(t[k] / sc) + 128, 0, 255).astype(np.uint8)
    special[k] = ((q.astype(np.float32) - np.float32(128)) * sc).astype(np.float32)
    named("nc1.weight_quantized", q)
    named("nc1.weight_scale", np.array(sc, dtype=np.float32))
    named("nc1.weight_zero_point", np.array(128, dtype=np.uint8))
    named("decoder.generator.noise_convs.1.bias", t["decoder.generator.noise_convs.1.bias"])
    nodes.append(OX.node_bytes("ConvInteger", ["xq", "nc1.weight_quantized", "xzp", "nc1.weight_zero_point"], ["ci"],
                               name="/decoder/generator/noise_convs.1/Conv_quant"))
    # 6. DequantizeLinear with block_size along axis 1 feeding a Conv
    k = "decoder.generator.conv_post.weight"
    a = t[k]  # [22, 128, 7]
    blk = 32
    nb = a.shape[1] // blk
    scb = (np.abs(a.reshape(a.shape[0], nb, blk, a.shape[2])).max(axis=2) / 127.0).astype(np.float32)  # [22, nb, 7]
    scb[scb == 0] = 1.0
    qb = np.clip(np.round(a / np.repeat(scb, blk, axis=1)), -127, 127).astype(np.int8)
    special[k] = (qb.astype(np.float32) * np.repeat(scb, blk, axis=1)).astype(np.float32)
    named("onnx::DQ_7", qb)
    named("onnx::DQ_7_s", scb)
    nodes.append(OX.node_bytes("DequantizeLinear", ["onnx::DQ_7", "onnx::DQ_7_s"], ["dq7"], name="/decoder/generator/conv_post/DequantizeLinear",
                               attrs={"axis": 1, "block_size": blk}))
    named("decoder.generator.conv_post.bias", t["decoder.generator.conv_post.bias"])
    nodes.append(OX.node_bytes("Conv", ["x", "dq7", "decoder.generator.conv_post.bias"], ["o7"], name="/decoder/generator/conv_post/Conv"))
    # 7. bfloat16 raw, double, float16 in int32_data, 'module.' and 'kmodel.' prefixes
    k = "predictor.F0_proj.weight"
    bf = (t[k].view(np.uint32) >> 16).astype(np.uint16)
    special[k] = (bf.astype(np.uint32) << 16).view(np.float32)
    named(k, bf, data_type=OX.BFLOAT16)
    k = "predictor.N_proj.weight"
    special[k] = t[k]
    named(k, t[k].astype(np.float64), data_type=OX.DOUBLE)
    k = "decoder.N_conv.weight"
    h = t[k].astype(np.float16)
    special[k] = h.astype(np.float32)
    inits.append(b"".join(OX._vi(1, int(d)) for d in h.shape) + OX._vi(2, OX.FLOAT16) + OX._ld(8, k.encode())
                 + OX._ld(5, b"".join(OX._wv(int(x)) for x in h.view(np.uint16).reshape(-1))))  # packed int32_data
    k = "decoder.F0_conv.weight"
    special[k] = t[k]
    named("module." + k, t[k])
    k = "text_encoder.embedding.weight"
    special[k] = t[k]
    named("kmodel." + k, t[k])
    done = set(special) | {"bert_encoder.bias", "predictor.duration_proj.linear_layer.bias", "bert.encoder.embedding_hidden_mapping_in.bias",
                           "decoder.asr_res.0.bias", "decoder.generator.noise_convs.1.bias", "decoder.generator.conv_post.bias"}
    for name, a in t.items():
        if name not in done:
            named(name, a)
    src = str(tmp_path / "exotic.onnx")
    with open(src, "wb") as f:
        f.write(OX.model_bytes(nodes, inits))
    py, cpp = src + ".py.kxw", src + ".cpp.kxw"
    I.import_onnx(src, py)
    hk.import_onnx(src, cpp)
    tp, tc = W.read_blob(py), W.read_blob(cpp)
    for name, want in special.items():
        np.testing.assert_array_equal(np.asarray(tp[name]), want.reshape(np.asarray(tp[name]).shape), err_msg=f"python: {name}")
        np.testing.assert_array_equal(np.asarray(tc[name]), want.reshape(np.asarray(tc[name]).shape), err_msg=f"c++: {name}")
    assert open(py, "rb").read() == open(cpp, "rb").read()
