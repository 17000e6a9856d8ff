"""Weight importer: upstream Kokoro-82M checkpoints -> the KXHIPW01 blob read by kx_create.

SURVEY.md §8f rank 4.  The reference never parses weights itself: it hands `onnx/model.onnx` of
onnx-community/Kokoro-82M-v1.0-ONNX to ONNX Runtime (/root/reference/kokorox/src/utils/hf_cache.rs:8-10,
128-158).  The same parameters are published upstream as a PyTorch checkpoint (hexgrad/Kokoro-82M
`kokoro-v1_0.pth`): a dict with one state dict per top-level module
    {"bert": {...}, "bert_encoder": {...}, "predictor": {...}, "text_encoder": {...}, "decoder": {...}}
whose keys may carry a DataParallel "module." prefix and whose convolutions are weight-normalised
(`weight_g`, `weight_v`).  This module flattens that, folds weight-norm
    w = g * v / ||v||   (norm over every dim except 0, torch.nn.utils.weight_norm default dim=0)
drops tensors the forward never reads (ALBERT pooler, position_ids, InstanceNorm affine placeholders, LSTM
flat-weights duplicates), checks every shape against kokorox_amd.weights.tensor_spec() and writes the blob.

ONNX files (`model.onnx` and its fp16 / int8 / 4-bit siblings, hf_cache.rs:135-144) are read with the
package-free protobuf walker in onnx_lite.py: `import_onnx` recovers the state-dict names from the graph -
initialisers that kept their parameter names are taken directly; weights the exporter renamed
("onnx::MatMul_123": transposed Linear weights, folded weight-norm convs, LSTM W/R/B) are named after the
module path in the node name ("/decoder/generator/resblocks.0/convs1.0/Conv") or after the named bias they
are added to; ONNX LSTM gate order iofc is turned back into PyTorch's ifgo; float16 / bfloat16 tensors are
widened and quantised weights (DequantizeLinear, MatMulInteger / ConvInteger with the onnxruntime
`_quantized/_scale/_zero_point` naming, com.microsoft MatMulNBits) are DE-quantised to f32 - the forward
stays f32-class, so a quantised file reproduces its weights, not ONNX Runtime's integer arithmetic.

No network and no real checkpoint exist in the build environment, so the tests exercise both paths on
synthetic files written here in the upstream / exporter style (tests/test_host_cpu.py,
tests/test_importer_onnx.py); the first run on a real file must also confirm the tensor names and shapes
assumed from SURVEY Appendix A (the importer lists what it could not place).
"""
from __future__ import annotations

import json
import struct
from collections import OrderedDict
from typing import Dict, Mapping  # noqa: F401

import numpy as np

from . import weights as W

_IGNORED_SUFFIXES = (
    "position_ids", "pooler.weight", "pooler.bias",      # ALBERT buffers / unused pooler
    ".norm.weight", ".norm.bias",                        # AdaIN1d InstanceNorm(affine=True) placeholders
    ".norm.running_mean", ".norm.running_var", ".norm.num_batches_tracked",
)


def _to_numpy(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t.astype(np.float32, copy=False)
    return t.detach().cpu().float().numpy()  # torch tensor


def flatten_checkpoint(ckpt: Mapping) -> "OrderedDict[str, np.ndarray]":
    """{"module": {"module.key": tensor}} or an already flat mapping -> flat {"module.key": array}."""
    flat: "OrderedDict[str, np.ndarray]" = OrderedDict()
    nested = all(isinstance(v, Mapping) for v in ckpt.values())
    if nested:
        for top, sd in ckpt.items():
            for k, v in sd.items():
                k = k[len("module."):] if k.startswith("module.") else k
                flat[f"{top}.{k}"] = _to_numpy(v)
    else:
        for k, v in ckpt.items():
            k = k[len("module."):] if k.startswith("module.") else k
            flat[k] = _to_numpy(v)
    return flat


def fold_weight_norm(flat: Mapping[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """Replace every (x.weight_g, x.weight_v) pair — also the parametrizations.weight.original0/1 spelling —
    by x.weight = g * v / ||v|| with the norm over all dims but 0."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    pairs = {}
    for k, v in flat.items():
        for g_suf, v_suf in ((".weight_g", ".weight_v"),
                             (".parametrizations.weight.original0", ".parametrizations.weight.original1")):
            if k.endswith(g_suf):
                pairs.setdefault(k[: -len(g_suf)], {})["g"] = v
                break
            if k.endswith(v_suf):
                pairs.setdefault(k[: -len(v_suf)], {})["v"] = v
                break
        else:
            out[k] = v
    for base, gv in pairs.items():
        if "g" not in gv or "v" not in gv:
            raise ValueError(f"incomplete weight-norm pair for {base}")
        v = gv["v"].astype(np.float64)
        g = gv["g"].astype(np.float64)
        # summed in index order (cumsum is strictly sequential; .sum() is pairwise): the in-library reader
        # (csrc/onnx_import.cpp) adds the squares in the same order, so both write the same bytes
        ss = np.cumsum(v.reshape(v.shape[0], -1) ** 2, axis=1)[:, -1]
        norm = np.sqrt(ss).reshape((-1,) + (1,) * (v.ndim - 1))
        out[base + ".weight"] = (g * v / norm).astype(np.float32)
    return out


def to_blob_tensors(ckpt: Mapping, strict: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Upstream checkpoint -> ordered tensors exactly as kokorox_amd.weights.tensor_spec() lists them."""
    flat = fold_weight_norm(flatten_checkpoint(ckpt))
    spec = W.tensor_spec()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    missing = []
    for name, (shape, _rule) in spec.items():
        a = flat.get(name)
        if a is None:
            missing.append(name)
            continue
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {tuple(a.shape)} != expected {tuple(shape)}")
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    if missing:
        raise KeyError(f"{len(missing)} tensors missing from the checkpoint, e.g. {missing[:5]}")
    if strict:
        extra = [k for k in flat if k not in spec and not k.endswith(_IGNORED_SUFFIXES)]
        if extra:
            raise KeyError(f"{len(extra)} unexpected tensors in the checkpoint, e.g. {extra[:5]}")
    return out


def read_safetensors(path: str) -> Dict[str, np.ndarray]:
    """Minimal safetensors reader (8-byte header length, JSON header, raw little-endian data), F32/F16/BF16."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n))
        base = 8 + n
        out = {}
        for k, meta in header.items():
            if k == "__metadata__":
                continue
            lo, hi = meta["data_offsets"]
            f.seek(base + lo)
            raw = f.read(hi - lo)
            dt = meta["dtype"]
            if dt == "F32":
                a = np.frombuffer(raw, dtype="<f4")
            elif dt == "F16":
                a = np.frombuffer(raw, dtype="<f2").astype(np.float32)
            elif dt == "BF16":
                a = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
            else:
                raise ValueError(f"{k}: unsupported safetensors dtype {dt}")
            out[k] = a.reshape(meta["shape"]).astype(np.float32)
    return out


# ---- ONNX ----------------------------------------------------------------------------------------------------
def _module_path(node_name: str) -> str:
    """'/decoder/generator/resblocks.0/convs1.0/Conv_1' -> 'decoder.generator.resblocks.0.convs1.0'."""
    parts = [p for p in node_name.strip("/").split("/") if p]
    return ".".join(parts[:-1]) if len(parts) > 1 else ""


def _lstm_gates_onnx_to_torch(a: np.ndarray, hid: int) -> np.ndarray:
    """Rows in ONNX order [i o f c] -> PyTorch order [i f g o] (g = ONNX's c)."""
    i, o, f, c = (a[k * hid:(k + 1) * hid] for k in range(4))
    return np.concatenate([i, f, c, o], axis=0)


def _dequant(x: np.ndarray, scale: np.ndarray, zp, axis: int = 1, block: int = 0) -> np.ndarray:
    x = x.astype(np.float32)
    scale = np.asarray(scale, dtype=np.float32)
    zp = np.zeros_like(scale) if zp is None else np.asarray(zp).astype(np.float32)
    if scale.ndim == 0 or scale.size == 1:
        return (x - zp.reshape(())) * scale.reshape(())
    if block and scale.shape != ():  # blocked along `axis` (DequantizeLinear-21)
        rep = np.repeat(scale, block, axis=axis)
        rz = np.repeat(zp, block, axis=axis)
        sl = tuple(slice(0, d) for d in x.shape)
        return (x - rz[sl]) * rep[sl]
    shape = [1] * x.ndim
    shape[axis if axis >= 0 else x.ndim + axis] = -1
    return (x - zp.reshape(shape)) * scale.reshape(shape)


def _matmul_nbits(node, const) -> np.ndarray:
    """com.microsoft MatMulNBits: B uint8 [N][K/block][block*bits/8], scales [N * K/block], optional packed zero points
    (default 2^(bits-1)); returns the [N, K] float weight (= PyTorch Linear.weight)."""
    K, N = int(node.attrs["K"]), int(node.attrs["N"])
    bits, bs = int(node.attrs.get("bits", 4)), int(node.attrs["block_size"])
    if bits != 4:
        raise ValueError(f"{node.name}: MatMulNBits with bits={bits} is not supported")
    nb = (K + bs - 1) // bs
    B = const(node.inputs[1], raw=True).reshape(N, nb, bs // 2)
    q = np.empty((N, nb, bs), dtype=np.float32)
    q[..., 0::2] = B & 0x0F
    q[..., 1::2] = B >> 4
    scales = const(node.inputs[2], raw=True).astype(np.float32).reshape(N, nb, 1)
    if len(node.inputs) > 3 and node.inputs[3]:
        zraw = const(node.inputs[3], raw=True).reshape(N, -1)
        z = np.empty((N, zraw.shape[1] * 2), dtype=np.float32)
        z[:, 0::2] = zraw & 0x0F
        z[:, 1::2] = zraw >> 4
        zp = z[:, :nb].reshape(N, nb, 1)
    else:
        zp = np.float32(8.0)
    return ((q - zp) * scales).reshape(N, nb * bs)[:, :K]


def onnx_to_state_dict(path: str):
    """model.onnx -> (flat {state-dict name: f32 array}, report).  The names are the upstream KModel names (with
    weight_g / weight_v pairs where the exporter kept them); `report` lists initialisers that were not placed."""
    from . import onnx_lite as OX
    g = OX.read_graph(path)
    spec = W.tensor_spec()
    inits = g.initializers
    derived: Dict[str, np.ndarray] = {}  # tensors computed from initialisers by Cast / Transpose / DequantizeLinear
    used = set()

    def const(name: str, raw: bool = False):
        if name in derived:
            return derived[name]
        t = inits.get(name)
        if t is None:
            return None
        used.add(name)
        return t.array if raw else np.asarray(t.array, dtype=np.float32)

    # constants the graph derives from initialisers before they reach a compute node
    for n in g.nodes:
        if not n.outputs:
            continue
        if n.op_type == "DequantizeLinear" and n.inputs[0] in inits:
            x = const(n.inputs[0], raw=True)
            zp = const(n.inputs[2], raw=True) if len(n.inputs) > 2 and n.inputs[2] else None
            derived[n.outputs[0]] = _dequant(x, const(n.inputs[1]), zp, int(n.attrs.get("axis", 1)),
                                             int(n.attrs.get("block_size", 0)))
        elif n.op_type == "Cast" and (n.inputs[0] in inits or n.inputs[0] in derived):
            derived[n.outputs[0]] = const(n.inputs[0])
        elif n.op_type == "Transpose" and (n.inputs[0] in inits or n.inputs[0] in derived):
            a = const(n.inputs[0])
            derived[n.outputs[0]] = np.transpose(a, n.attrs.get("perm") or list(range(a.ndim))[::-1])

    flat: "OrderedDict[str, np.ndarray]" = OrderedDict()
    wn_suffixes = (".weight_g", ".weight_v", ".parametrizations.weight.original0", ".parametrizations.weight.original1")

    def canonical(name: str):
        """The spec (or weight-norm / ignorable) name an initialiser name stands for, tolerating one extra or one
        missing leading component (exporters wrap the model: 'kmodel.bert...' / 'encoder...')."""
        cands = [name] + ([name.split(".", 1)[1]] if "." in name else [])
        for c in cands:
            base = c
            for suf in wn_suffixes:
                if c.endswith(suf):
                    base = c[: -len(suf)] + ".weight"
            if base in spec or c.endswith(_IGNORED_SUFFIXES):
                return c
        tails = [k for k in spec if k.endswith("." + name)]
        return tails[0] if len(tails) == 1 else None

    def place(name: str, a: np.ndarray, why: str):
        target = spec.get(name)
        if target is not None:
            shape = tuple(target[0])
            if a.ndim == len(shape) + 1 and 1 in a.shape:  # 1-D convs exported as 2-D ones
                a = a.reshape([d for i, d in enumerate(a.shape) if not (d == 1 and i == 2)])
            if tuple(a.shape) != shape and a.size == int(np.prod(shape)):
                a = a.reshape(shape)  # e.g. alpha [ch] vs [1, ch, 1]
        if name not in flat:
            flat[name] = np.ascontiguousarray(a, dtype=np.float32)

    def spec_name(path: str, leaf: str):
        """module path from a node name -> full state-dict name, also when the path lacks leading components or carries
        up to two extra ones (exporters wrap the model: '/kmodel/bert/...', '/model/kmodel/bert/...')."""
        full = f"{path}.{leaf}" if path else leaf
        cand = full
        for _ in range(3):
            if cand in spec:
                return cand
            if "." not in cand:
                break
            cand = cand.split(".", 1)[1]
        tails = [k for k in spec if k.endswith("." + full)]
        return tails[0] if len(tails) == 1 else None

    # 1. initialisers that kept a parameter name
    for name in list(inits):
        c = canonical(name)
        if c is not None and inits[name].data_type in (OX.FLOAT, OX.FLOAT16, OX.BFLOAT16, OX.DOUBLE):
            place(c, const(name), "name")

    # 2. weights the exporter renamed: find them through the nodes that consume them
    consumers: Dict[str, list] = {}
    for n in g.nodes:
        for i in n.inputs:
            consumers.setdefault(i, []).append(n)
    for n in g.nodes:
        path = _module_path(n.name)
        op = n.op_type
        if op in ("Conv", "ConvTranspose", "ConvInteger"):
            wname = n.inputs[1]
            if canonical(wname) is None:
                a = const(wname)
                if op == "ConvInteger" and a is not None:
                    zp = const(n.inputs[3], raw=True) if len(n.inputs) > 3 and n.inputs[3] else None
                    sc = const(wname.replace("_quantized", "_scale"))
                    if sc is None:
                        raise ValueError(f"{n.name}: no '{wname.replace('_quantized', '_scale')}' initialiser for ConvInteger")
                    a = _dequant(const(wname, raw=True), sc, zp, axis=0)
                tgt = spec_name(path, "weight")
                if a is not None and tgt:
                    place(tgt, a, "conv node")
            # (ConvInteger's third input is the activation zero point, not a bias)
            if op != "ConvInteger" and len(n.inputs) > 2 and n.inputs[2] and canonical(n.inputs[2]) is None:
                b = const(n.inputs[2])
                tgt = spec_name(path, "bias")
                if b is not None and tgt:
                    place(tgt, b, "conv node")
        elif op in ("MatMul", "MatMulInteger", "MatMulNBits", "Gemm"):
            wt = None
            if op == "MatMulNBits":
                wt = _matmul_nbits(n, const)
            elif op == "MatMulInteger":
                bq = const(n.inputs[1], raw=True)
                if bq is not None:
                    zp = const(n.inputs[3], raw=True) if len(n.inputs) > 3 and n.inputs[3] else None
                    sc = const(n.inputs[1].replace("_quantized", "_scale"))
                    if sc is None:
                        raise ValueError(f"{n.name}: no scale initialiser beside {n.inputs[1]}")
                    wt = _dequant(bq, sc, zp, axis=1).T
            elif op == "Gemm":
                b = const(n.inputs[1])
                if b is not None and b.ndim == 2:
                    wt = b if int(n.attrs.get("transB", 0)) else b.T
            else:
                b = const(n.inputs[1])
                if b is not None and b.ndim == 2:
                    wt = b.T  # MatMul(x, W^T): the exporter stores Linear.weight transposed
            if wt is None:
                continue
            tgt = spec_name(path, "weight")
            bias_arr, bias_name = None, None
            if op == "Gemm" and len(n.inputs) > 2 and n.inputs[2]:
                bias_arr, bias_name = const(n.inputs[2]), n.inputs[2]
            else:  # the Add that follows a MatMul carries the (named) bias
                outs = [n.outputs[0]]
                for hop in range(3):  # MatMulInteger: Cast / Mul sit between the product and the Add
                    nxt = [c for o in outs for c in consumers.get(o, [])]
                    adds = [c for c in nxt if c.op_type == "Add"]
                    if adds:
                        other = [i for i in adds[0].inputs if i not in outs]
                        if other and const(other[0]) is not None and const(other[0]).ndim == 1:
                            bias_arr, bias_name = const(other[0]), other[0]
                        break
                    outs = [c.outputs[0] for c in nxt if c.op_type in ("Cast", "Mul") and c.outputs]
                    if not outs:
                        break
            if tgt is None and bias_name is not None:
                c = canonical(bias_name)
                if c is not None and c.endswith(".bias"):
                    tgt = c[: -len(".bias")] + ".weight"
            if tgt and tgt in spec and wt.size == int(np.prod(spec[tgt][0])):
                place(tgt, wt, "matmul node")
                if bias_arr is not None and canonical(bias_name) is None:
                    place(tgt[: -len(".weight")] + ".bias", bias_arr, "matmul node")
        elif op == "LSTM":
            Wi, R = const(n.inputs[1]), const(n.inputs[2])
            B = const(n.inputs[3]) if len(n.inputs) > 3 and n.inputs[3] else None
            hid = int(n.attrs.get("hidden_size", Wi.shape[1] // 4))
            base = spec_name(path, "weight_ih_l0")
            if base is None:
                continue
            base = base[: -len(".weight_ih_l0")]
            for d in range(Wi.shape[0]):
                suf = "" if d == 0 else "_reverse"
                place(f"{base}.weight_ih_l0{suf}", _lstm_gates_onnx_to_torch(Wi[d], hid), "lstm node")
                place(f"{base}.weight_hh_l0{suf}", _lstm_gates_onnx_to_torch(R[d], hid), "lstm node")
                if B is not None:
                    place(f"{base}.bias_ih_l0{suf}", _lstm_gates_onnx_to_torch(B[d][: 4 * hid], hid), "lstm node")
                    place(f"{base}.bias_hh_l0{suf}", _lstm_gates_onnx_to_torch(B[d][4 * hid:], hid), "lstm node")
    report = {"initializers": len(inits), "placed": len(flat),
              "unused_initializers": sorted(k for k in inits if k not in used and inits[k].array.size > 16)[:50]}
    return flat, report


def import_onnx(src: str, dst: str, strict: bool = False) -> str:
    """`model.onnx` (fp32, fp16 or a quantised variant) -> KXHIPW01 blob at dst."""
    flat, report = onnx_to_state_dict(src)
    try:
        tensors = to_blob_tensors(flat, strict=strict)
    except KeyError as e:
        raise KeyError(f"{e.args[0]}; initialisers the importer could not place: {report['unused_initializers'][:10]}") from None
    W.write_blob(dst, tensors)
    return dst


def import_checkpoint(src: str, dst: str, strict: bool = True) -> str:
    """`.onnx`, `.pth`/`.pt` (torch.load) or `.safetensors` -> KXHIPW01 blob at dst."""
    if src.endswith(".onnx"):
        return import_onnx(src, dst, strict=False)
    if src.endswith(".safetensors"):
        ckpt = read_safetensors(src)
    else:
        import torch
        ckpt = torch.load(src, map_location="cpu", weights_only=True)
        if "net" in ckpt and isinstance(ckpt["net"], Mapping):
            ckpt = ckpt["net"]
    W.write_blob(dst, to_blob_tensors(ckpt, strict=strict))
    return dst


if __name__ == "__main__":  # python -m kokorox_amd.importer model.onnx|kokoro-v1_0.pth|model.safetensors kokoro.kxw
    import sys
    print(import_checkpoint(sys.argv[1], sys.argv[2]))
