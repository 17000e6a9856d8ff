"""Kokoro-82M tensor table, weight-blob container and seeded synthetic weights.

The reference never holds weights in-tree: `OrtBase::load_model`
(/root/reference/kokorox/src/onn/ort_base.rs:14-39) hands a path to ONNX Runtime, and
the file comes from Hugging Face at run time (kokorox/src/utils/hf_cache.rs:8-10,149).
There is no network here, so the model is exercised with *synthetic* weights of the real
architecture, scaled so that activations, durations, F0 and the
output waveform stay in realistic ranges.

Blob format "KXHIPW01" (little endian), consumed by kokorox_amd/csrc/weights.cpp and by
the oracle's own reader:

    0   char[8]  magic "KXHIPW01"
    8   u32      n_tensors
    12  u32      table_bytes (n_tensors * 128)
    16  u64      data_offset (256-aligned)
    24  u64      total_bytes
    32..63       reserved (zero)
    64  table: per tensor 128 bytes = char name[88]; u32 dtype (0=f32); u32 ndim;
               u32 dims[4]; u64 offset (from file start, 256-aligned); u64 nbytes
    data_offset.. raw tensors, canonical PyTorch layouts (weight-norm already folded)

Tensor names follow the upstream `KModel` state-dict naming so that a future importer
 is a rename plus weight-norm fold.
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"KXHIPW01"
ENTRY_BYTES = 128
NAME_BYTES = 88

# ---- hyper-parameters -----------------------------------------
N_TOKEN = 178
HIDDEN = 512
STYLE = 128
MAX_DUR = 50
BERT_H = 768
BERT_E = 128
BERT_HEADS = 12
BERT_FF = 2048
BERT_LAYERS = 12
BERT_MAXPOS = 512
N_FFT = 20
HOP = 5
UPS_RATES = (10, 6)
UPS_KERNELS = (20, 12)
UPS_CH0 = 512
RES_KERNELS = (3, 7, 11)
RES_DIL = (1, 3, 5)
HARMONICS = 9
SAMPLE_RATE = 24000
SAMPLES_PER_FRAME = 600


def _lstm(spec, prefix, n_in, hid=HIDDEN // 2):
    for suf in ("", "_reverse"):
        spec[f"{prefix}.weight_ih_l0{suf}"] = ((4 * hid, n_in), ("lstm", hid))
        spec[f"{prefix}.weight_hh_l0{suf}"] = ((4 * hid, hid), ("lstm", hid))
        spec[f"{prefix}.bias_ih_l0{suf}"] = ((4 * hid,), ("lstm", hid))
        spec[f"{prefix}.bias_hh_l0{suf}"] = ((4 * hid,), ("lstm", hid))


def _adain_fc(spec, prefix, ch):
    spec[f"{prefix}.fc.weight"] = ((2 * ch, STYLE), ("normal", 0.25))
    spec[f"{prefix}.fc.bias"] = ((2 * ch,), ("normal", 0.05))


def _conv(spec, prefix, cout, cin, k, bias=True, gain=1.0):
    spec[f"{prefix}.weight"] = ((cout, cin, k), ("fan_in", gain, cin * k))
    if bias:
        spec[f"{prefix}.bias"] = ((cout,), ("normal", 0.02))


def _adain_resblk(spec, prefix, din, dout, upsample=False):
    _conv(spec, f"{prefix}.conv1", dout, din, 3, gain=1.3)
    _conv(spec, f"{prefix}.conv2", dout, dout, 3, gain=1.3)
    _adain_fc(spec, f"{prefix}.norm1", din)
    _adain_fc(spec, f"{prefix}.norm2", dout)
    if din != dout:
        _conv(spec, f"{prefix}.conv1x1", dout, din, 1, bias=False)
    if upsample:
        # depth-wise ConvTranspose1d(din, din, k3, s2, groups=din): weight [din,1,3]
        spec[f"{prefix}.pool.weight"] = ((din, 1, 3), ("normal", 0.6))
        spec[f"{prefix}.pool.bias"] = ((din,), ("normal", 0.02))


def _adain_resblock1(spec, prefix, ch, k):
    for i in range(3):
        _conv(spec, f"{prefix}.convs1.{i}", ch, ch, k, gain=0.9)
        _conv(spec, f"{prefix}.convs2.{i}", ch, ch, k, gain=0.9)
        _adain_fc(spec, f"{prefix}.adain1.{i}", ch)
        _adain_fc(spec, f"{prefix}.adain2.{i}", ch)
        spec[f"{prefix}.alpha1.{i}"] = ((1, ch, 1), ("uniform", 0.5, 2.0))
        spec[f"{prefix}.alpha2.{i}"] = ((1, ch, 1), ("uniform", 0.5, 2.0))


def tensor_spec() -> "OrderedDict[str, tuple]":
    """name -> (shape, init rule). Order is the blob order."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    # ---- PL-BERT (ALBERT, one shared layer) -------------------------------------------
    e = "bert.embeddings"
    s[f"{e}.word_embeddings.weight"] = ((N_TOKEN, BERT_E), ("normal", 0.5))
    s[f"{e}.position_embeddings.weight"] = ((BERT_MAXPOS, BERT_E), ("normal", 0.3))
    s[f"{e}.token_type_embeddings.weight"] = ((2, BERT_E), ("normal", 0.1))
    s[f"{e}.LayerNorm.weight"] = ((BERT_E,), ("ones_jitter", 0.05))
    s[f"{e}.LayerNorm.bias"] = ((BERT_E,), ("normal", 0.02))
    s["bert.encoder.embedding_hidden_mapping_in.weight"] = ((BERT_H, BERT_E), ("fan_in", 1.0, BERT_E))
    s["bert.encoder.embedding_hidden_mapping_in.bias"] = ((BERT_H,), ("normal", 0.02))
    l = "bert.encoder.albert_layer_groups.0.albert_layers.0"
    for nm in ("query", "key", "value", "dense"):
        s[f"{l}.attention.{nm}.weight"] = ((BERT_H, BERT_H), ("fan_in", 1.0, BERT_H))
        s[f"{l}.attention.{nm}.bias"] = ((BERT_H,), ("normal", 0.02))
    s[f"{l}.attention.LayerNorm.weight"] = ((BERT_H,), ("ones_jitter", 0.05))
    s[f"{l}.attention.LayerNorm.bias"] = ((BERT_H,), ("normal", 0.02))
    s[f"{l}.ffn.weight"] = ((BERT_FF, BERT_H), ("fan_in", 1.0, BERT_H))
    s[f"{l}.ffn.bias"] = ((BERT_FF,), ("normal", 0.02))
    s[f"{l}.ffn_output.weight"] = ((BERT_H, BERT_FF), ("fan_in", 1.0, BERT_FF))
    s[f"{l}.ffn_output.bias"] = ((BERT_H,), ("normal", 0.02))
    s[f"{l}.full_layer_layer_norm.weight"] = ((BERT_H,), ("ones_jitter", 0.05))
    s[f"{l}.full_layer_layer_norm.bias"] = ((BERT_H,), ("normal", 0.02))
    s["bert_encoder.weight"] = ((HIDDEN, BERT_H), ("fan_in", 1.0, BERT_H))
    s["bert_encoder.bias"] = ((HIDDEN,), ("normal", 0.02))
    # ---- ProsodyPredictor ---------------------------------------------------------------
    for i in range(3):
        _lstm(s, f"predictor.text_encoder.lstms.{2 * i}", HIDDEN + STYLE)
        s[f"predictor.text_encoder.lstms.{2 * i + 1}.fc.weight"] = ((2 * HIDDEN, STYLE), ("normal", 0.25))
        s[f"predictor.text_encoder.lstms.{2 * i + 1}.fc.bias"] = ((2 * HIDDEN,), ("normal", 0.05))
    _lstm(s, "predictor.lstm", HIDDEN + STYLE)
    s["predictor.duration_proj.linear_layer.weight"] = ((MAX_DUR, HIDDEN), ("normal", 0.25))
    s["predictor.duration_proj.linear_layer.bias"] = ((MAX_DUR,), ("const_jitter", -3.2, 0.3))
    _lstm(s, "predictor.shared", HIDDEN + STYLE)
    for br in ("F0", "N"):
        _adain_resblk(s, f"predictor.{br}.0", HIDDEN, HIDDEN)
        _adain_resblk(s, f"predictor.{br}.1", HIDDEN, HIDDEN // 2, upsample=True)
        _adain_resblk(s, f"predictor.{br}.2", HIDDEN // 2, HIDDEN // 2)
    # F0 in Hz: mean ~120, spread ~110, so utterances mix voiced (>10 Hz) and unvoiced frames
    s["predictor.F0_proj.weight"] = ((1, HIDDEN // 2, 1), ("normal", 7.0))
    s["predictor.F0_proj.bias"] = ((1,), ("const_jitter", 120.0, 0.0))
    s["predictor.N_proj.weight"] = ((1, HIDDEN // 2, 1), ("normal", 0.08))
    s["predictor.N_proj.bias"] = ((1,), ("const_jitter", 0.5, 0.0))
    # ---- TextEncoder ---------------------------------------------------------------------
    s["text_encoder.embedding.weight"] = ((N_TOKEN, HIDDEN), ("normal", 1.0))
    for i in range(3):
        _conv(s, f"text_encoder.cnn.{i}.0", HIDDEN, HIDDEN, 5, gain=1.3)
        s[f"text_encoder.cnn.{i}.1.gamma"] = ((HIDDEN,), ("ones_jitter", 0.05))
        s[f"text_encoder.cnn.{i}.1.beta"] = ((HIDDEN,), ("normal", 0.05))
    _lstm(s, "text_encoder.lstm", HIDDEN)
    # ---- Decoder --------------------------------------------------------------------------
    _adain_resblk(s, "decoder.encode", HIDDEN + 2, 1024)
    for i in range(3):
        _adain_resblk(s, f"decoder.decode.{i}", 1024 + 2 + 64, 1024)
    _adain_resblk(s, "decoder.decode.3", 1024 + 2 + 64, 512, upsample=True)
    s["decoder.F0_conv.weight"] = ((1, 1, 3), ("normal", 0.004))
    s["decoder.F0_conv.bias"] = ((1,), ("normal", 0.02))
    s["decoder.N_conv.weight"] = ((1, 1, 3), ("normal", 0.5))
    s["decoder.N_conv.bias"] = ((1,), ("normal", 0.02))
    _conv(s, "decoder.asr_res.0", 64, HIDDEN, 1)
    g = "decoder.generator"
    s[f"{g}.m_source.l_linear.weight"] = ((1, HARMONICS), ("normal", 1.2))
    s[f"{g}.m_source.l_linear.bias"] = ((1,), ("normal", 0.02))
    _conv(s, f"{g}.noise_convs.0", 256, N_FFT + 2, 12)
    _conv(s, f"{g}.noise_convs.1", 128, N_FFT + 2, 1)
    _adain_resblock1(s, f"{g}.noise_res.0", 256, 7)
    _adain_resblock1(s, f"{g}.noise_res.1", 128, 11)
    # ConvTranspose1d weight layout is [Cin, Cout, k]; each output sees Cin*k/stride taps
    s[f"{g}.ups.0.weight"] = ((512, 256, 20), ("fan_in", 1.0, 512 * 2))
    s[f"{g}.ups.0.bias"] = ((256,), ("normal", 0.02))
    s[f"{g}.ups.1.weight"] = ((256, 128, 12), ("fan_in", 1.0, 256 * 2))
    s[f"{g}.ups.1.bias"] = ((128,), ("normal", 0.02))
    for i, ch in enumerate((256, 128)):
        for j, k in enumerate(RES_KERNELS):
            _adain_resblock1(s, f"{g}.resblocks.{i * 3 + j}", ch, k)
    # head: log-magnitude (ch 0..10) centred at -0.6, phase pre-activation (ch 11..21) wide
    s[f"{g}.conv_post.weight"] = ((N_FFT + 2, 128, 7), ("fan_in", 0.25, 128 * 7))
    s[f"{g}.conv_post.bias"] = ((N_FFT + 2,), ("post_bias",))
    return s


def n_params() -> int:
    return int(sum(int(np.prod(shape)) for shape, _ in tensor_spec().values()))


def _init(rng: np.random.Generator, shape, rule) -> np.ndarray:
    kind = rule[0]
    n = int(np.prod(shape))
    if kind == "normal":
        a = rng.standard_normal(n, dtype=np.float32) * np.float32(rule[1])
    elif kind == "fan_in":
        a = rng.standard_normal(n, dtype=np.float32) * np.float32(rule[1] / np.sqrt(rule[2]))
    elif kind == "uniform":
        a = rng.random(n, dtype=np.float32) * np.float32(rule[2] - rule[1]) + np.float32(rule[1])
    elif kind == "lstm":
        b = 1.0 / np.sqrt(rule[1])
        a = (rng.random(n, dtype=np.float32) * np.float32(2 * b) - np.float32(b))
    elif kind == "ones_jitter":
        a = np.float32(1.0) + rng.standard_normal(n, dtype=np.float32) * np.float32(rule[1])
    elif kind == "const_jitter":
        a = np.float32(rule[1]) + rng.standard_normal(n, dtype=np.float32) * np.float32(rule[2])
    elif kind == "post_bias":
        a = np.zeros(n, np.float32)
        a[: N_FFT // 2 + 1] = -0.6
    else:  # pragma: no cover
        raise ValueError(kind)
    return a.astype(np.float32).reshape(shape)


def _layout(spec):
    """(table, data_offset, total_bytes) with 256-byte aligned tensor offsets."""
    n = len(spec)
    data_off = (64 + n * ENTRY_BYTES + 255) // 256 * 256
    off = data_off
    table = []
    for name, (shape, _) in spec.items():
        nbytes = int(np.prod(shape)) * 4
        table.append((name, shape, off, nbytes))
        off = (off + nbytes + 255) // 256 * 256
    return table, data_off, off


def write_blob(path: str, tensors: "OrderedDict[str, np.ndarray]") -> None:
    """Write an ordered name->float32 array dict in KXHIPW01 format."""
    spec = OrderedDict((k, (tuple(v.shape), None)) for k, v in tensors.items())
    table, data_off, total = _layout(spec)
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as f:
        f.truncate(total)
    mm = np.memmap(tmp, dtype=np.uint8, mode="r+", shape=(total,))
    hdr = MAGIC + struct.pack("<IIQQ", len(table), len(table) * ENTRY_BYTES, data_off, total)
    mm[: len(hdr)] = np.frombuffer(hdr, np.uint8)
    for i, (name, shape, off, nbytes) in enumerate(table):
        nb = name.encode()
        assert len(nb) < NAME_BYTES and len(shape) <= 4, name
        dims = list(shape) + [0] * (4 - len(shape))
        ent = nb.ljust(NAME_BYTES, b"\0") + struct.pack("<II4IQQ", 0, len(shape), *dims, off, nbytes)
        assert len(ent) == ENTRY_BYTES
        mm[64 + i * ENTRY_BYTES: 64 + (i + 1) * ENTRY_BYTES] = np.frombuffer(ent, np.uint8)
        a = np.ascontiguousarray(tensors[name], dtype=np.float32)
        mm[off: off + nbytes] = a.view(np.uint8).reshape(-1)
    mm.flush()
    del mm
    os.replace(tmp, path)


def synthetic_tensors(seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (shape, rule) in tensor_spec().items():
        out[name] = _init(rng, shape, rule)
    return out


def default_blob_path(seed: int = 1234) -> str:
    root = os.environ.get("KOKOROX_AMD_CACHE") or os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build")
    return os.path.join(root, f"kokoro82m_synth_seed{seed}.kxw")


def ensure_synthetic_blob(path: str | None = None, seed: int = 1234) -> str:
    """Create the seeded synthetic weight file if it is not there yet; return its path."""
    path = path or default_blob_path(seed)
    if os.path.exists(path):
        return path
    os.makedirs(os.path.dirname(path), exist_ok=True)
    write_blob(path, synthetic_tensors(seed))
    return path


def read_blob(path: str) -> "OrderedDict[str, np.ndarray]":
    """Memory-mapped read (used by host-side tools; the oracle has its own reader)."""
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    if bytes(mm[:8]) != MAGIC:
        raise ValueError(f"{path}: not a KXHIPW01 weight blob")
    n, _tb, _doff, total = struct.unpack("<IIQQ", bytes(mm[8:32]))
    if total != mm.shape[0]:
        raise ValueError(f"{path}: truncated ({mm.shape[0]} of {total} bytes)")
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(n):
        ent = bytes(mm[64 + i * ENTRY_BYTES: 64 + (i + 1) * ENTRY_BYTES])
        name = ent[:NAME_BYTES].split(b"\0", 1)[0].decode()
        _dt, nd, d0, d1, d2, d3, off, nbytes = struct.unpack("<II4IQQ", ent[NAME_BYTES:])
        shape = (d0, d1, d2, d3)[:nd]
        out[name] = mm[off: off + nbytes].view(np.float32).reshape(shape)
    return out


def synthetic_voices(n_voices: int = 54, seed: int = 1) -> np.ndarray:
    """Voice table stand-in: [n,511,1,256] ~ N(0,0.1^2), row 510 zero.

    Geometry follows the reference's combined voices file: 510 rows per voice padded to
    511 (/root/reference/kokorox/src/utils/hf_cache.rs:284-309)."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n_voices, 511, 1, 256), dtype=np.float32) * np.float32(0.1)
    v[:, 510] = 0
    return v
