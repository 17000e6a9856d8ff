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

No network and no real checkpoint exist in the build environment, so the tests exercise it on a synthetic
upstream-style checkpoint (tests/test_host_cpu.py); ONNX initialisers are not handled yet (no `onnx`
package here) — export them to a state dict first.
"""
from __future__ import annotations

import json
import struct
from collections import OrderedDict
from typing import Dict, Mapping

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
        norm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1)).reshape((-1,) + (1,) * (v.ndim - 1))
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


def import_checkpoint(src: str, dst: str, strict: bool = True) -> str:
    """`.pth`/`.pt` (torch.load) or `.safetensors` -> KXHIPW01 blob at dst."""
    if src.endswith(".safetensors"):
        ckpt = read_safetensors(src)
    else:
        import torch
        ckpt = torch.load(src, map_location="cpu", weights_only=True)
        if "net" in ckpt and isinstance(ckpt["net"], Mapping):
            ckpt = ckpt["net"]
    W.write_blob(dst, to_blob_tensors(ckpt, strict=strict))
    return dst


if __name__ == "__main__":  # python -m kokorox_amd.importer kokoro-v1_0.pth kokoro.kxw
    import sys
    print(import_checkpoint(sys.argv[1], sys.argv[2]))
