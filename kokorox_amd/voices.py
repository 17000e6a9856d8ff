"""Caller-side producers of the hot path's inputs, mirrored from the reference so that
tests and the harness build `infer` arguments exactly as `TTSKoko` does.

Reference (paths under /root/reference/kokorox/src/tts/):
  * `TTSKoko::mix_styles`   koko.rs:1255-1306  — style row lookup and the "a.4+b.5" blend
  * `TTSKoko::load_voices`  koko.rs:1308-1334  — NPZ -> {name: [511][1][256]}
  * `tokenize_with_variant` tokenize.rs:35-67  — char -> id, unknown chars dropped
  * padding                 koko.rs:1161-1175  — optional id-30 prefix, 0 at both ends
  * chunk loop              koko.rs:947-1191   — one forward per <= 500-token chunk, waveforms appended
These stay on the host (north_star: "the voice-style mixer ... stay identical"); the GPU
only ever sees ids and one 256-float row per utterance.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

VOICE_ROWS = 511


def load_voices(path: str) -> Dict[str, np.ndarray]:
    """NPZ of per-voice arrays [rows<=511, 1, 256] -> zero-padded [511, 1, 256] float32."""
    out: Dict[str, np.ndarray] = {}
    with np.load(path) as npz:
        for name in npz.files:
            a = np.asarray(npz[name], dtype=np.float32)
            t = np.zeros((VOICE_ROWS, 1, 256), dtype=np.float32)
            r = min(a.shape[0], VOICE_ROWS)
            t[:r, : a.shape[1], : a.shape[2]] = a[:r, :1, :256]
            out[name] = t
    return out


def mix_styles(styles: Dict[str, np.ndarray], style_name: str, tokens_len: int) -> List[List[float]]:
    """[[256 floats]] for `style_name` at row `tokens_len` (token count BEFORE the 0 padding).

    A mix "a.4+b.5" is sum(row * (weight * 0.1)) with NO normalisation; parts without '.' or
    with a non-numeric weight are silently skipped; an unknown voice is an error."""
    if "+" not in style_name:
        if style_name not in styles:
            raise KeyError(f"can not found from styles_map: {style_name}")
        return [np.asarray(styles[style_name][tokens_len][0], dtype=np.float32).tolist()]
    names, portions = [], []
    for part in style_name.split("+"):
        if "." not in part:
            continue
        name, portion = part.split(".", 1)
        try:
            p = np.float32(float(portion))
        except ValueError:
            continue
        if name not in styles:
            raise KeyError(f"Voice '{name}' not found in available voices")
        names.append(name)
        portions.append(np.float32(p * np.float32(0.1)))
    if not names:
        raise ValueError(f"Invalid voice mix format '{style_name}'. Use format: voice1.weight+voice2.weight "
                         "(e.g., jf_alpha.4+am_echo.6)")
    blended = np.zeros(256, dtype=np.float32)
    for name, p in zip(names, portions):
        blended = (blended + np.asarray(styles[name][tokens_len][0], dtype=np.float32) * p).astype(np.float32)
    return [blended.tolist()]


def tokenize(phonemes: str, symbols: Sequence[str]) -> List[int]:
    """char -> index in the symbol table; unknown characters are dropped (tokenize.rs:43-55).

    A symbol that occurs twice (the apostrophe, ids 174 and 176 in the v1.0 table) maps to
    its LAST index here; the reference's HashMap makes that choice order-dependent."""
    table = {c: i for i, c in enumerate(symbols)}
    return [table[c] for c in phonemes if c in table]


def pad_tokens(tokens: Sequence[int], initial_silence: int = 0) -> List[List[int]]:
    """koko.rs:1161-1175: optional id-30 prefix, then 0 at both ends, as a batch of one."""
    t = [30] * int(initial_silence) + list(tokens)
    return [[0] + t + [0]]


def tts_chunks(model, styles: Dict[str, np.ndarray], style_name: str, chunk_tokens: Sequence[Sequence[int]],
               speed: float = 1.0, initial_silence: int = 0, seed: int = 0) -> np.ndarray:
    """Mirror of the chunk loop of `TTSKoko::tts_raw_audio` (koko.rs:947-1191) from the point
    where a chunk has been tokenised: per chunk the optional id-30 prefix, `mix_styles(style, tokens.len())` (the
    style row depends on the chunk's own token count, koko.rs:1165), zero padding at both ends, one forward, and
    the waveforms appended one after the other with no cross-fade (koko.rs:1177-1180).

    The reference runs the chunks one by one through its single `Mutex<Session>`; chunks are independent, so here
    they form ONE batched forward (`model.infer_batch`), which on the GPU gives the same samples as chunk-by-chunk
    calls (utterance b of a batch draws its noise from key (seed, b): tests/test_gpu_forward.py).  An empty chunk
    list gives an empty waveform; an empty chunk is an error, as `OrtKoko::infer` would index past the end
    (ort_koko.rs:56)."""
    toks, rows = [], []
    for ch in chunk_tokens:
        t = [30] * int(initial_silence) + [int(v) for v in ch]
        if not t:
            raise ValueError("tts_chunks: empty chunk")
        rows.append(mix_styles(styles, style_name, len(t))[0])
        toks.append([0] + t + [0])
    if not toks:
        return np.zeros(0, dtype=np.float32)
    outs = model.infer_batch(toks, np.asarray(rows, dtype=np.float32), [float(speed)], seed=seed)
    return np.concatenate([np.asarray(o, dtype=np.float32) for o in outs])
