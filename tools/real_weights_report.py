"""Real-weights readiness report: run once on the first real blob.

    python tools/real_weights_report.py [weights.kxw] [--apply]

For a handful of inputs it runs the forward in both contraction modes (kx_set_conv_mode: 0 = f32 MFMA, exact f32
products; 1 = f16x3 split) with KX_FLAG_TAPS and prints
  * per tap: max |f16x3 - f32| relative to the tap's magnitude (the A/B DESIGN.md §3 names as the first check), and
  * per conv layer: absmax and rms of its input after the AdaIN affine (kx_diag_*), flagging what the split cannot
    carry: |x| > 6e4 (clamped at 65504) and rms < 1e-3 (low half lost to the f16 subnormal quantum), with the
    power-of-two pre-scale that brings the layer back to rms ~ 1 (kx_set_act_prescale); for the 7- / 11-tap convs, whose cross
    terms the default f16f8 mode carries on e4m3 images, also the narrower window of those (2^-6 .. 448).
With --apply the suggested pre-scales are set and the A/B is repeated.  Without a path the seeded synthetic blob is used
(every layer is then in range: the report is the tool's own smoke test).
"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402

# Taps up to the F0 / N curves.  The generator sits behind the F0 -> phase edge: a last-bit change of F0
# decorrelates the harmonic source, so generator taps of two fp32-class runs differ by O(1) whatever the arithmetic;
# the generator's convs are covered by the per-layer input diagnostics below and by the kernel-level tests.
TAPS = ["bert.out", "d_en", "dur.lstm", "pred.shared", "pred.F0", "pred.N", "text_enc.out", "dec.encode", "dec.decode.0",
        "dec.decode.1", "dec.decode.2", "dec.decode.3"]


def synthetic_inputs(n_phonemes, seed):
    rng = np.random.default_rng(seed)
    return [0] + rng.integers(1, 178, size=n_phonemes).tolist() + [0]


def ab(model, cases):
    worst = {}
    default_mode = model.get_conv_mode()
    for ids, style in cases:
        taps = {}
        for mode in (0, 1):
            model.set_conv_mode(mode)
            model.infer([ids], [style], 1.0, seed=2, flags=hk.KX_FLAG_TAPS)
            taps[mode] = {t: model.tap(t, 0) for t in TAPS}
        for t in TAPS:
            a, b = taps[0][t], taps[1][t]
            if a.shape != b.shape:  # durations differ between the modes: everything downstream is incomparable
                worst[t] = float("inf")
                continue
            d = float(np.abs(a - b).max() / max(1.0, float(np.abs(a).max())))
            worst[t] = max(worst.get(t, 0.0), d)
    model.set_conv_mode(default_mode)
    return worst


def main():
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    apply = "--apply" in sys.argv
    blob = args[0] if args else W.ensure_synthetic_blob()
    model = hk.HipKoko.new(blob)
    voices = W.synthetic_voices(2)
    cases = [(synthetic_inputs(n, 40 + n), list(voices[i % 2, n, 0])) for i, n in enumerate((12, 40, 128))]
    model.set_pinned_durations([3, 3, 3, 4])  # same frame counts in both modes: every tap stays comparable
    print(f"blob: {blob}")
    print("== A/B f32 MFMA vs f16x3 split, per tap (max |d| / max(1, |tap|max)), worst of 3 utterances")
    w = ab(model, cases)
    for t in TAPS:
        flag = "  <-- above 5e-5" if w[t] > 5e-5 else ""
        print(f"  {t:18s} {w[t]:.3e}{flag}")
    model.diag_enable(True)
    for ids, style in cases:
        model.infer([ids], [style], 1.0, seed=2)
    recs = model.diag_records()
    model.diag_enable(False)
    layers = {}
    for name, rows, cin, k, sh, amax, rms, cnt in recs:
        e = layers.setdefault(name, [rows, cin, k, sh, 0.0, np.inf, 0])
        e[4] = max(e[4], amax)
        e[5] = min(e[5], rms)
        e[6] += 1
    print(f"== conv inputs after the AdaIN affine: {len(layers)} layers, {len(recs)} launches")
    print(f"  {'layer':58s} {'rows':>5s} {'Cin':>5s} {'k':>3s} {'absmax':>10s} {'min rms':>10s}  note")
    suggest = {}
    for name, (rows, cin, k, sh, amax, rms, n) in sorted(layers.items()):
        note = ""
        if amax * 2.0 ** sh > 6e4:
            note = "absmax > 6e4: clamped by the f16 split"
            suggest[name] = sh - int(np.ceil(np.log2(amax * 2.0 ** sh / 1e4)))
        elif 0 < rms * 2.0 ** sh < 1e-3:
            e = int(np.round(-np.log2(rms)))
            note = f"rms < 1e-3: low halves lost; suggest kx_set_act_prescale(\"{name}\", {e})"
            suggest[name] = e
        elif k in (7, 11) and 0 < rms * 2.0 ** sh < 2.0 ** -5:
            # the default mode's 7- / 11-tap convs carry their cross terms on e4m3 images (normal range 2^-6 .. 448): below it the
            # images lose bits and a product keeps less than the mode's 2^-17 (never less than one f16 MFMA's 2^-12)
            e = int(np.round(-np.log2(rms)))
            note = f"f16f8: rms below e4m3's normal range; suggest kx_set_act_prescale(\"{name}\", {e})"
            suggest[name] = e
        elif k in (7, 11) and amax * 2.0 ** sh > 448.0:
            note = ("f16f8: |x| > 448: the cross terms of those elements are clamped (their products keep 2^-12, the rest 2^-17); "
                    "KOKOROX_CONV=f16x3 carries them fully")
        print(f"  {name:58s} {rows:5d} {cin:5d} {k:3d} {amax:10.3e} {rms:10.3e}  {note}")
    if not suggest:
        print("every layer is inside the range the split carries exactly (1e-3 <= rms, absmax <= 6e4)")
    elif apply:
        for name, e in suggest.items():
            model.set_act_prescale(name, e)
        print(f"== applied {len(suggest)} pre-scales; A/B again")
        w2 = ab(model, cases)
        for t in TAPS:
            print(f"  {t:18s} {w[t]:.3e} -> {w2[t]:.3e}")
    model.close()


if __name__ == "__main__":
    main()
