"""Generate tests/golden/ fixtures from the CPU oracle (oracle/kokoro_ref.py).

The reference holds no golden vector for the forward pass ("parity
unpinned"), so the waveform fixtures pin the ORACLE against regressions and give the GPU
tests vectors that do not need the oracle to run.  Input-side fixtures that the reference
DOES hold are copied as data: the token rows of kokorox/src/tts/tokenize.rs:120-129 and
kokorox/src/onn/ort_koko.rs:46, and the v1.0 symbol table of kokorox/src/tts/vocab.rs:7-10.

    python tools/make_golden.py        # rewrites tests/golden/*.npz, *.json
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kokorox_amd import weights as W  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# kokorox/src/tts/vocab.rs:7-10 (data): pad + punctuation + letters + IPA letters, id = index
PAD = "$"
PUNCT = ";:,.!?¡¿—…\"«»“” "
LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
IPA = ("ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁǂǃˈˌːˑʼʴʰʱʲʷˠˤ˞↓↑→↗↘'̩'ᵻ")

REFERENCE_TOKEN_ROWS = {
    # tokenize.rs:120-122
    "Hello!": [24, 47, 54, 54, 57, 5],
    # tokenize.rs:124-129
    "$həlˈoʊ, wˈɜːld!$": [0, 50, 83, 54, 156, 57, 135, 3, 16, 65, 156, 87, 158, 54, 46, 5, 0],
}
# ort_koko.rs:46 (comment holding a sample `tokens` row)
ORT_KOKO_SAMPLE_ROW = [0, 56, 51, 142, 156, 69, 63, 3, 16, 61, 4, 16, 156, 51, 4, 16, 62, 77, 156, 51, 86, 5, 0]

TAPS = ["d_en", "dur.lstm", "pred.F0", "pred.N", "text_enc.out", "gen.har_source", "audio"]


def main():
    os.makedirs(GOLD, exist_ok=True)
    symbols = list(PAD + PUNCT + LETTERS + IPA)
    assert len(symbols) == 178, len(symbols)
    with open(os.path.join(GOLD, "inputs_reference.json"), "w", encoding="utf-8") as f:
        json.dump({"source": "byteowlz/kokorox: vocab.rs:7-10, tokenize.rs:120-129, ort_koko.rs:46",
                   "symbols": symbols, "token_rows": REFERENCE_TOKEN_ROWS,
                   "ort_koko_sample_row": ORT_KOKO_SAMPLE_ROW}, f, ensure_ascii=False, indent=1)

    blob = W.ensure_synthetic_blob()
    o = R.KokoroOracle(blob)
    voices = W.synthetic_voices(2)
    cases = {
        # the reference's own token rows as inputs (style row index = tokens before padding)
        "hello_world": (np.array(REFERENCE_TOKEN_ROWS["$həlˈoʊ, wˈɜːld!$"], np.int64), voices[0, 15, 0], 1.0),
        "ort_sample_row": (np.array(ORT_KOKO_SAMPLE_ROW, np.int64), voices[1, 21, 0], 1.25),
    }
    for name, (ids, style, speed) in cases.items():
        taps = {}
        audio, dur = o.forward(ids, style, speed, seed=2, utt=0, taps=taps)
        out = {"ids": ids, "style": style.astype(np.float32), "speed": np.float32(speed), "seed": np.int64(2),
               "pred_dur": dur.numpy().astype(np.int64), "weights_seed": np.int64(1234)}
        for t in TAPS:
            out["tap:" + t] = taps[t].numpy().astype(np.float32)
        # the same utterance through the other STFT pair (torch.stft / torch.istft semantics), F0 / N curves pinned
        o_t = R.KokoroOracle(blob, stft_variant="torch")
        audio_t, _ = o_t.forward(ids, style, speed, seed=2, utt=0, f0_override=taps["pred.F0"][0].numpy(),
                                 n_override=taps["pred.N"][0].numpy())
        out["tap:audio_torch_stft"] = audio_t.numpy().astype(np.float32)[None]
        np.savez_compressed(os.path.join(GOLD, f"forward_{name}.npz"), **out)
        print(name, "F =", int(dur.sum()), "samples =", audio.shape[0], "max|a| =", float(audio.abs().max()))
    # Philox / Box-Muller stream (first values), pins the noise definition shared with the HIP kernel
    z = R.gauss_noise(seed=0x1234567890ABCDEF, utt=3, n_samples=64)
    np.savez_compressed(os.path.join(GOLD, "noise_stream.npz"), z=z, seed=np.uint64(0x1234567890ABCDEF), utt=np.int64(3))


if __name__ == "__main__":
    main()
