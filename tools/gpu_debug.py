"""GPU bring-up report: kernel hooks vs torch, then every tap of a small batch vs the oracle.

Prints a table instead of asserting, so one gpurun call localises the first diverging stage.
Usage (on a GPU box): python tools/gpu_debug.py [--tokens 16,11] [--out gpurun_out/debug.txt]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402


def rel(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return float(d.max()), float(d.max() / (np.abs(b).max() + 1e-30))


def ops_report(log):
    rng = np.random.default_rng(0)
    cases = [
        # B, Cin, Cout, L, k, stride, pad, dil
        (2, 16, 32, 70, 3, 1, 1, 1),
        (1, 24, 130, 300, 7, 1, 9, 3),
        (2, 128, 128, 517, 11, 1, 25, 5),
        (1, 22, 256, 601, 12, 6, 3, 1),
        (2, 1, 1, 40, 3, 2, 1, 1),
        (1, 640, 50, 33, 1, 1, 0, 1),
        (1, 1090, 64, 45, 3, 1, 1, 1),
    ]
    for (B, Cin, Cout, L, k, s, p, d) in cases:
        x = rng.standard_normal((B, Cin, L), dtype=np.float32)
        w = rng.standard_normal((Cout, Cin, k), dtype=np.float32) / np.sqrt(Cin * k)
        b = rng.standard_normal(Cout, dtype=np.float32)
        y = hk.conv1d(x, w, b, stride=s, pad=p, dil=d)
        ref = torch.nn.functional.conv1d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                                         torch.from_numpy(b).double(), stride=s, padding=p, dilation=d).numpy()
        log(f"conv1d B{B} {Cin}->{Cout} L{L} k{k} s{s} p{p} d{d}: max|d| {rel(y, ref)[0]:.3e}")
    # fused AdaIN + snake / leaky input transform
    B, Cin, Cout, L, k = 2, 40, 48, 200, 7
    x = rng.standard_normal((B, Cin, L), dtype=np.float32)
    w = rng.standard_normal((Cout, Cin, k), dtype=np.float32) / np.sqrt(Cin * k)
    norm = rng.standard_normal((3, B, Cin), dtype=np.float32)
    alpha = rng.random(Cin, dtype=np.float32) + 0.5
    for act, name in ((2, "snake"), (1, "leaky")):
        y = hk.conv1d(x, w, None, pad=3, act=act, slope=0.2, alpha=alpha, norm=norm)
        xt = (torch.from_numpy(x).double() - torch.from_numpy(norm[0]).double()[:, :, None]) * \
            torch.from_numpy(norm[1]).double()[:, :, None] + torch.from_numpy(norm[2]).double()[:, :, None]
        if act == 2:
            a = torch.from_numpy(alpha).double()[None, :, None]
            xt = xt + (1 / a) * torch.sin(a * xt) ** 2
        else:
            xt = torch.nn.functional.leaky_relu(xt, 0.2)
        ref = torch.nn.functional.conv1d(xt, torch.from_numpy(w).double(), padding=3).numpy()
        log(f"conv1d norm+{name}: max|d| {rel(y, ref)[0]:.3e}")
    # transposed (polyphase)
    for (Cin, Cout, L, s) in ((16, 24, 37, 10), (256, 128, 50, 6)):
        k = 2 * s
        x = rng.standard_normal((2, Cin, L), dtype=np.float32)
        w = rng.standard_normal((Cin, Cout, k), dtype=np.float32) / np.sqrt(Cin * 2)
        b = rng.standard_normal(Cout, dtype=np.float32)
        y = hk.conv1d(x, w, b, stride=s, pad=(k - s) // 2, transposed=True)
        ref = torch.nn.functional.conv_transpose1d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                                                   torch.from_numpy(b).double(), stride=s, padding=(k - s) // 2).numpy()
        log(f"convT {Cin}->{Cout} L{L} s{s}: max|d| {rel(y, ref)[0]:.3e}")
    # LSTM
    torch.manual_seed(0)
    m = torch.nn.LSTM(72, 256, 1, batch_first=True, bidirectional=True)
    x = torch.randn(2, 19, 72)
    with torch.no_grad():
        ref = m(x)[0].numpy()
    ps = [getattr(m, n + suf).detach().numpy() for suf in ("", "_reverse")
          for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    y = hk.lstm(x.numpy(), ps)
    log(f"lstm: max|d| {rel(y, ref)[0]:.3e}")


def source_report(log, oracle):
    rng = np.random.default_rng(3)
    F2 = 40
    f0 = (rng.standard_normal((2, F2)) * 120 + 110).astype(np.float32)
    lw = oracle.w["decoder.generator.m_source.l_linear.weight"].numpy()
    lb = float(oracle.w["decoder.generator.m_source.l_linear.bias"][0])
    y = hk.harmonic_source(f0, lw, lb, seed=7, utt_base=5)
    for b in range(2):
        ref = oracle.source(torch.from_numpy(f0[b]), 7, 5 + b, 1.0, {}).numpy()
        d = np.abs(y[b] - ref)
        log(f"harmonic source utt{b}: max|d| {d.max():.3e}  n(|d|>1e-5) {int((d > 1e-5).sum())} of {d.size}")
    y0 = hk.harmonic_source(f0, lw, lb, seed=7, utt_base=5, noise_off=True)
    ref = oracle.source(torch.from_numpy(f0[0]), 7, 5, 0.0, {}).numpy()
    log(f"harmonic source noise-off: max|d| {np.abs(y0[0] - ref).max():.3e}")


def taps_report(log, oracle, blob, token_counts):
    B = len(token_counts)
    ids_all = [R.synthetic_inputs(1, n, seed=10 + i)[0] for i, n in enumerate(token_counts)]
    voices = W.synthetic_voices(4)
    styles = [voices[i % 4, n, 0] for i, n in enumerate(token_counts)]
    t = time.time()
    m = hk.HipKoko.new(blob)
    log(f"kx_create: {time.time() - t:.2f} s")
    t = time.time()
    outs = m.infer_batch([list(x) for x in ids_all], styles, [1.0], seed=2, flags=hk.KX_FLAG_TAPS)
    log(f"kx_infer (taps on): {time.time() - t:.2f} s; samples {[len(o) for o in outs]}")
    for b in range(B):
        taps = {}
        audio, dur = oracle.forward(ids_all[b], styles[b], 1.0, seed=2, utt=b, taps=taps)
        log(f"--- utterance {b}: T={len(ids_all[b])} oracle F={int(dur.sum())} gpu samples={len(outs[b])}")
        if len(outs[b]) != audio.shape[0]:
            log("   LENGTH MISMATCH (durations differ)")
        for name, ref in taps.items():
            if name == "dur.duration":
                continue
            try:
                g = m.tap(name, b)
            except hk.KokoroxHipError as e:
                log(f"   {name:18s} (no tap: {e})")
                continue
            refn = ref.float().numpy()
            if g.shape != refn.shape:
                log(f"   {name:18s} SHAPE gpu {g.shape} oracle {refn.shape}")
                continue
            mx, rl = rel(g, refn)
            log(f"   {name:18s} max|d| {mx:.3e}  rel {rl:.3e}  scale {np.abs(refn).max():.3f}")
        # teacher-forced comparison: oracle re-run with the GPU's F0 / N curves
        f0 = m.tap("pred.F0", b)[0]
        n = m.tap("pred.N", b)[0]
        taps2 = {}
        audio2, _ = oracle.forward(ids_all[b], styles[b], 1.0, seed=2, utt=b, taps=taps2, f0_override=f0, n_override=n)
        har_g = m.tap("gen.har", b)
        taps3 = {}
        oracle.forward(ids_all[b], styles[b], 1.0, seed=2, utt=b, taps=taps3, f0_override=f0, n_override=n, har_override=har_g)
        for name in ("gen.x_source.0", "gen.stage.0", "gen.stage.1", "gen.conv_post", "audio"):
            mx, rl = rel(m.tap(name, b), taps3[name].float().numpy())
            log(f"   [F0+STFT pinned] {name:15s} max|d| {mx:.3e}  rel {rl:.3e}")
        for name in ("gen.har_source", "gen.har", "gen.x_source.0", "gen.ups.0", "gen.stage.0", "gen.x_source.1",
                     "gen.ups.1", "gen.stage.1", "gen.conv_post", "audio"):
            g = m.tap(name, b)
            refn = taps2[name].float().numpy()
            if g.shape != refn.shape:
                log(f"   [F0 pinned] {name:15s} SHAPE gpu {g.shape} oracle {refn.shape}")
                continue
            mx, rl = rel(g, refn)
            log(f"   [F0 pinned] {name:15s} max|d| {mx:.3e}  rel {rl:.3e}")
            if name == "gen.har" and mx > 1.0:
                hs = torch.from_numpy(m.tap("gen.har_source", b)[0])
                xp = torch.nn.functional.pad(hs[None, None], (10, 10), mode="replicate")
                re = torch.nn.functional.conv1d(xp, oracle.fwd_re, stride=5)[0].numpy()
                im = torch.nn.functional.conv1d(xp, oracle.fwd_im, stride=5)[0].numpy()
                for (c, f) in np.argwhere(np.abs(g - refn) > 1.0)[:12]:
                    k = c - 11
                    log(f"      flip at bin {k} frame {f}/{g.shape[1]}: gpu {g[c, f]:+.6f} oracle {refn[c, f]:+.6f}  "
                        f"cpu re {re[k, f]:+.3e} im {im[k, f]:+.3e} signbit(im) {bool(np.signbit(im[k, f]))}  "
                        f"gpu mag {g[k, f]:.3e} window {hs[max(0, 5 * f - 10): 5 * f + 10].numpy().round(4).tolist()}")
    m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="16,11")
    ap.add_argument("--out", default="")
    ap.add_argument("--skip-ops", action="store_true")
    a = ap.parse_args()
    fh = open(a.out, "w") if a.out else None

    def log(s):
        print(s, flush=True)
        if fh:
            fh.write(s + "\n")
            fh.flush()

    blob = W.ensure_synthetic_blob()
    oracle = R.KokoroOracle(blob)
    log(hk.load_library().kx_version().decode())
    if not a.skip_ops:
        ops_report(log)
        source_report(log, oracle)
    taps_report(log, oracle, blob, [int(x) for x in a.tokens.split(",")])


if __name__ == "__main__":
    main()
