"""Per-tap and waveform error of the opt-in reduced-precision mode (KOKOROX_CONV=f16 / kx_set_conv_mode(4)) against the
CPU oracle, next to the default f16x3 mode, under the usual protocol (F0 / N curves and source STFT pinned).
    python tools/reduced_precision_report.py > profiles/r03_reduced_precision.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402

BACK = ["dec.encode", "dec.decode.0", "dec.decode.3", "gen.x_source.0", "gen.ups.0", "gen.stage.0", "gen.x_source.1", "gen.ups.1",
        "gen.stage.1", "gen.conv_post", "audio"]


def main():
    blob = W.ensure_synthetic_blob()
    m = hk.HipKoko.new(blob)
    o = R.KokoroOracle(blob)
    print("# reduced-precision mode (one f16 MFMA per product in the decoder / generator direct-A convs) vs the default f16x3")
    print("# mode, each against the CPU oracle; synthetic weights; F0 / N curves and source STFT of the GPU run pinned in the oracle")
    print("# utterance tokens mode tap max|ref| max|d| max|d|/max|ref|")
    for n_tok, seed0 in ((24, 500), (60, 501), (128, 502)):
        ids = R.synthetic_inputs(1, n_tok, seed=seed0)[0]
        style = W.synthetic_voices(1)[0, n_tok, 0]
        for mode, name in ((1, "f16x3"), (4, "f16")):
            m.set_conv_mode(mode)
            out = m.infer([list(ids)], [list(style)], 1.0, seed=2, flags=hk.KX_FLAG_TAPS)
            f0, n_c, har = m.tap("pred.F0", 0), m.tap("pred.N", 0)[0], m.tap("gen.har", 0)
            taps = {}
            audio, dur = o.forward(ids, style, 1.0, seed=2, utt=0, taps=taps, f0_override=f0[0], n_override=n_c, har_override=har)
            assert out.shape[0] == 600 * int(dur.sum())
            for t in BACK:
                ref = taps[t].numpy() if t != "audio" else audio.numpy()[None]
                got = m.tap(t, 0) if t != "audio" else out[None]
                d = float(np.abs(got - ref).max())
                print(f"{seed0} {n_tok:4d} {name:6s} {t:16s} {np.abs(ref).max():10.4f} {d:10.3e} {d / max(np.abs(ref).max(), 1e-30):10.3e}")
    m.close()


if __name__ == "__main__":
    main()
