"""CPU: the ONNX-initialiser importer (kokorox_amd.importer.import_onnx + onnx_lite) on synthetic model.onnx files
serialised here in the style of the PyTorch exporter - the only kind of weight file the reference loads
(/root/reference/kokorox/src/utils/hf_cache.rs:8-10, variants hf_cache.rs:135-144):

  * Linear layers as MatMul with a TRANSPOSED, anonymously named weight ("onnx::MatMul_17") + Add of the named bias;
  * convolutions three ways: weight_g / weight_v initialisers (weight norm kept), a folded anonymous weight with a
    named bias, and an anonymous weight on a bias-less conv that only the node name identifies;
  * LSTMs as ONNX LSTM nodes (W / R / B stacked over directions, gate order i o f c);
  * fp16 storage; int8 (MatMulInteger naming convention, per-channel DequantizeLinear) and 4-bit (MatMulNBits) weights.
"""
import os

import numpy as np
import pytest

from kokorox_amd import importer as I
from kokorox_amd import onnx_lite as OX
from kokorox_amd import weights as W


def _torch_to_onnx_gates(a, hid=256):
    i, f, g, o = (a[k * hid:(k + 1) * hid] for k in range(4))
    return np.concatenate([i, o, f, g], axis=0)


def _dress_as_export(tensors, style="fp32", rng=None, pooler=768):
    """state-dict tensors -> (node bytes list, initializer bytes list) of an exporter-looking graph."""
    rng = rng or np.random.default_rng(0)
    nodes, inits = [], []
    uid = [0]

    def anon(kind):
        uid[0] += 1
        return f"onnx::{kind}_{uid[0]}"

    def add_init(name, a, quant=None):
        if style == "fp16" and a.dtype == np.float32:
            inits.append(OX.tensor_bytes(name, a.astype(np.float16)))
        else:
            inits.append(OX.tensor_bytes(name, a, raw=(uid[0] % 2 == 0)))  # both float_data and raw_data encodings

    spec = W.tensor_spec()
    done = set()
    # LSTMs first (four tensors per direction)
    for name in spec:
        if name.endswith(".weight_ih_l0") and name in tensors:  # (a caller may dress a subset of the model)
            base = name[: -len(".weight_ih_l0")]
            Wi = np.stack([_torch_to_onnx_gates(tensors[f"{base}.weight_ih_l0{s}"]) for s in ("", "_reverse")])
            R = np.stack([_torch_to_onnx_gates(tensors[f"{base}.weight_hh_l0{s}"]) for s in ("", "_reverse")])
            B = np.stack([np.concatenate([_torch_to_onnx_gates(tensors[f"{base}.bias_ih_l0{s}"]),
                                          _torch_to_onnx_gates(tensors[f"{base}.bias_hh_l0{s}"])]) for s in ("", "_reverse")])
            wn, rn, bn = anon("LSTM"), anon("LSTM"), anon("LSTM")
            for nm, a in ((wn, Wi), (rn, R), (bn, B)):
                add_init(nm, a.astype(np.float32))
            nodes.append(OX.node_bytes("LSTM", ["x", wn, rn, bn], ["y"], name="/" + base.replace(".", "/") + "/LSTM",
                                       attrs={"hidden_size": 256, "direction": "bidirectional"}))
            for s in ("", "_reverse"):
                for t in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                    done.add(f"{base}.{t}{s}")
    conv_style = 0
    for name, a in tensors.items():
        if name in done:
            continue
        base = name[: -len(".weight")] if name.endswith(".weight") else None
        is_linear = base is not None and a.ndim == 2 and "embeddings" not in name and "embedding.weight" not in name
        is_conv = base is not None and a.ndim == 3 and not name.endswith("pool.weight")
        path = "/" + (base or name).replace(".", "/")
        if is_linear:
            bias = tensors.get(base + ".bias")
            if style == "int8":
                scale = np.float32(np.abs(a).max() / 127.0)
                q = np.clip(np.round(a.T / scale) + 128, 0, 255).astype(np.uint8)
                qn = base + ".weight_quantized"
                add_init(qn, q)
                add_init(base + ".weight_scale", np.array(scale, dtype=np.float32))
                add_init(base + ".weight_zero_point", np.array(128, dtype=np.uint8))
                nodes.append(OX.node_bytes("MatMulInteger", ["xq", qn, "xzp", base + ".weight_zero_point"], [path + "/mmi"],
                                           name=path + "/MatMul_quant"))
                nodes.append(OX.node_bytes("Cast", [path + "/mmi"], [path + "/cast"], name=path + "/Cast", attrs={"to": 1}))
                nodes.append(OX.node_bytes("Mul", [path + "/cast", "xs"], [path + "/mm"], name=path + "/Mul"))
            elif style == "q4" and a.shape[1] % 32 == 0 and a.shape[0] >= 256:
                N, K = a.shape
                blk = a.reshape(N, K // 32, 32)
                sc = (np.abs(blk).max(axis=2, keepdims=True) / 7.0).astype(np.float32)
                sc[sc == 0] = 1.0
                q = np.clip(np.round(blk / sc) + 8, 0, 15).astype(np.uint8)
                packed = (q[..., 0::2] | (q[..., 1::2] << 4)).astype(np.uint8)
                bn_, sn_ = anon("MatMulNBits"), anon("MatMulNBits")
                add_init(bn_, packed)
                add_init(sn_, sc.reshape(-1))
                nodes.append(OX.node_bytes("MatMulNBits", ["x", bn_, sn_], [path + "/mm"], name=path + "/MatMul_Q4",
                                           attrs={"K": K, "N": N, "bits": 4, "block_size": 32}, domain="com.microsoft"))
            else:
                wn = anon("MatMul")
                add_init(wn, np.ascontiguousarray(a.T))
                nodes.append(OX.node_bytes("MatMul", ["x", wn], [path + "/mm"], name=path + "/MatMul"))
            if bias is not None:
                add_init(base + ".bias", bias)
                nodes.append(OX.node_bytes("Add", [base + ".bias", path + "/mm"], [path + "/out"], name=path + "/Add"))
                done.add(base + ".bias")
        elif is_conv:
            bias = tensors.get(base + ".bias")
            conv_style = (conv_style + 1) % 3
            if style == "int8" and bias is not None:
                scale = (np.abs(a).reshape(a.shape[0], -1).max(axis=1) / 127.0).astype(np.float32)
                scale[scale == 0] = 1.0
                q = np.clip(np.round(a / scale[:, None, None]), -127, 127).astype(np.int8)
                qn = anon("Conv")
                add_init(qn, q)
                add_init(qn + "_s", scale)
                add_init(qn + "_z", np.zeros(a.shape[0], dtype=np.int8))
                nodes.append(OX.node_bytes("DequantizeLinear", [qn, qn + "_s", qn + "_z"], [qn + "_dq"],
                                           name=path + "/DequantizeLinear", attrs={"axis": 0}))
                add_init(base + ".bias", bias)
                nodes.append(OX.node_bytes("Conv", ["x", qn + "_dq", base + ".bias"], [path + "/out"], name=path + "/Conv"))
                done.add(base + ".bias")
            elif conv_style == 0 and bias is not None and style == "fp32":  # weight norm kept: g and v are initialisers
                v = (a.astype(np.float64) * rng.uniform(0.5, 2.0, size=(a.shape[0], 1, 1))).astype(np.float32)
                gq = np.sqrt((a.astype(np.float64).reshape(a.shape[0], -1) ** 2).sum(1)).reshape(-1, 1, 1).astype(np.float32)
                add_init(base + ".weight_g", gq)
                add_init(base + ".weight_v", v)
                add_init(base + ".bias", bias)
                nodes.append(OX.node_bytes("Conv", ["x", path + "/normed", base + ".bias"], [path + "/out"], name=path + "/Conv"))
                done.add(base + ".bias")
            else:  # folded weight, anonymous; the bias (if any) keeps its name
                wn = anon("Conv")
                a4 = a[:, :, None, :] if conv_style == 1 else a  # some exporters emit 1-D convs as 2-D ones
                add_init(wn, a4)
                ins = ["x", wn]
                if bias is not None:
                    add_init(base + ".bias", bias)
                    ins.append(base + ".bias")
                    done.add(base + ".bias")
                nodes.append(OX.node_bytes("Conv", ins, [path + "/out"], name=path + "/Conv"))
        else:
            if name.endswith("pool.weight"):  # depth-wise ConvTranspose: anonymous weight + anonymous bias, node name only
                wn, bn_ = anon("ConvTranspose"), anon("ConvTranspose")
                add_init(wn, a)
                add_init(bn_, tensors[name[: -len(".weight")] + ".bias"])
                nodes.append(OX.node_bytes("ConvTranspose", ["x", wn, bn_], [path + "/out"],
                                           name="/" + name[: -len(".weight")].replace(".", "/") + "/ConvTranspose",
                                           attrs={"group": int(a.shape[0])}))
                done.add(name[: -len(".weight")] + ".bias")
            elif "alpha" in name:
                add_init(name, a.reshape(-1) if conv_style == 2 else a)  # [ch] or [1, ch, 1]
            else:
                add_init(name, a)
        done.add(name)
    # things a real file also holds and the importer must ignore
    add_init("bert.embeddings.position_ids", np.arange(512, dtype=np.int64)[None])
    add_init("bert.pooler.weight", np.zeros((pooler, pooler), np.float32))
    return nodes, inits


@pytest.fixture(scope="module")
def synth():
    return W.read_blob(W.ensure_synthetic_blob())


_CACHE = {}


def _roundtrip(synth, tmp_path, style):
    if style in _CACHE:  # (one export + import per style and test session)
        return _CACHE[style]
    _CACHE[style] = _roundtrip_uncached(synth, tmp_path, style)
    return _CACHE[style]


def _roundtrip_uncached(synth, tmp_path, style):
    nodes, inits = _dress_as_export({k: np.asarray(v) for k, v in synth.items()}, style)
    src = str(tmp_path / f"model_{style}.onnx")
    with open(src, "wb") as f:
        f.write(OX.model_bytes(nodes, inits))
    dst = str(tmp_path / f"model_{style}.kxw")
    assert I.import_checkpoint(src, dst) == dst
    return W.read_blob(dst)


def test_fp32_onnx_roundtrip(synth, tmp_path):
    got = _roundtrip(synth, tmp_path, "fp32")
    assert list(got.keys()) == list(synth.keys())
    for k, ref in synth.items():
        ref = np.asarray(ref)
        a = np.asarray(got[k])
        assert a.shape == ref.shape, k
        if ref.ndim == 3:  # possibly through a weight-norm fold: g * v / |v| in f64, then f32
            np.testing.assert_allclose(a, ref, rtol=2e-6, atol=1e-7, err_msg=k)
        else:
            np.testing.assert_array_equal(a, ref, err_msg=k)


def test_lstm_gate_order_is_converted(synth, tmp_path):
    """ONNX i o f c -> PyTorch i f g o, both directions, biases split into ih and hh halves."""
    got = _roundtrip(synth, tmp_path, "fp32")
    for k in ("predictor.lstm.weight_ih_l0_reverse", "text_encoder.lstm.bias_hh_l0", "predictor.shared.weight_hh_l0"):
        np.testing.assert_array_equal(np.asarray(got[k]), np.asarray(synth[k]), err_msg=k)
    a = np.arange(1024, dtype=np.float32)
    back = I._lstm_gates_onnx_to_torch(_torch_to_onnx_gates(a), 256)
    np.testing.assert_array_equal(back, a)


def test_fp16_onnx_roundtrip(synth, tmp_path):
    got = _roundtrip(synth, tmp_path, "fp16")
    for k, ref in synth.items():
        ref = np.asarray(ref)
        np.testing.assert_allclose(np.asarray(got[k]), ref.astype(np.float16).astype(np.float32), rtol=0, atol=0, err_msg=k)


@pytest.mark.parametrize("style,tol", [("int8", 1.0 / 127), ("q4", 1.0 / 7)])
def test_quantised_onnx_is_dequantised(synth, tmp_path, style, tol):
    """int8 (MatMulInteger naming, per-channel DequantizeLinear) and 4-bit MatMulNBits weights come back as f32 within
    half a quantisation step of the originals; everything that was not quantised is exact."""
    got = _roundtrip(synth, tmp_path, style)
    n_q = 0
    for k, ref in synth.items():
        ref = np.asarray(ref)
        a = np.asarray(got[k])
        assert a.shape == ref.shape, k
        if np.array_equal(a, ref):
            continue
        n_q += 1
        if ref.ndim == 3:
            step = np.abs(ref).reshape(ref.shape[0], -1).max(axis=1)[:, None, None] * tol
        elif style == "q4":
            step = (np.abs(ref.reshape(ref.shape[0], -1, 32)).max(axis=2, keepdims=True) * tol).repeat(32, axis=2).reshape(ref.shape)
        else:
            step = np.abs(ref).max() * tol
        assert np.all(np.abs(a - ref) <= 0.5001 * step + 1e-7), k
    assert n_q >= 10


def test_missing_tensor_is_reported(synth, tmp_path):
    nodes, inits = _dress_as_export({k: np.asarray(v) for k, v in synth.items()}, "fp32")
    # drop one LSTM node: its W / R / B initialisers are anonymous and can no longer be placed
    keep = [n for n in nodes if b"/predictor/lstm/LSTM" not in n]
    assert len(keep) == len(nodes) - 1
    src = str(tmp_path / "broken.onnx")
    with open(src, "wb") as f:
        f.write(OX.model_bytes(keep, inits))
    with pytest.raises(KeyError, match="predictor.lstm") as ei:
        I.import_onnx(src, str(tmp_path / "x.kxw"))
    assert "could not place" in str(ei.value) and "onnx::LSTM" in str(ei.value)
    with pytest.raises(ValueError, match="GraphProto"):
        bad = tmp_path / "bad.onnx"
        bad.write_bytes(b"\x08\x08")
        I.import_onnx(str(bad), str(tmp_path / "y.kxw"))


def test_command_line(synth, tmp_path):
    import subprocess
    import sys
    nodes, inits = _dress_as_export({k: np.asarray(v) for k, v in synth.items()}, "fp32")
    src = str(tmp_path / "model.onnx")
    with open(src, "wb") as f:
        f.write(OX.model_bytes(nodes, inits))
    dst = str(tmp_path / "out.kxw")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "kokorox_amd.importer", src, dst], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(dst) == os.path.getsize(W.ensure_synthetic_blob())
