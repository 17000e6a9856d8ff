"""GPU: the whole forward through kx_infer against the CPU oracle, the golden vectors and
size-independent properties.

Parity protocol (DESIGN.md "Parity"): every stage up to and including the predicted F0 / N curves
is compared directly.  The F0 curve then enters a ~1e5 rad phase accumulation, where one float32 ulp
of F0 decorrelates the harmonic source (the oracle's own fp32 and fp64 runs differ by >0.1 in the
waveform), so the harmonic source and its STFT are compared with that edge pinned (oracle re-run on the GPU's
F0 / N curves; STFT phases modulo 2*pi, because atan2's branch cut is a second discontinuity).  With
the STFT pinned as well, everything downstream must agree to |d| < 1e-4 (north_star tolerance).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_WAVE = 1e-4
REDUCED_TOL = 2e-3  # opt-in f16 mode: bound on max|d| / max(1, max|waveform|), measured (profiles/r03_reduced_precision.txt)
REDUCED_TOL_BF16 = 1.6e-2  # opt-in bf16 mode: 8 significant bits instead of 11
FRONT = ["bert.emb", "bert.layer0", "bert.out", "d_en", "dur_enc.0", "dur_enc.1", "dur_enc.2", "dur.lstm",
         "pred.shared", "pred.N", "text_enc.cnn", "text_enc.out", "dec.encode", "dec.decode.0", "dec.decode.1",
         "dec.decode.2", "dec.decode.3"]
BACK = ["gen.x_source.0", "gen.ups.0", "gen.stage.0", "gen.x_source.1", "gen.ups.1",
        "gen.stage.1", "gen.conv_post", "audio"]


def _wrapped_diff(got, ref):
    """STFT taps (11 magnitude rows, 11 phase rows): |got - ref| for the magnitudes; for the phases the difference
    modulo 2*pi, weighted by min(1, 50 * magnitude): atan2 of a bin whose magnitude is at rounding level is
    ill-conditioned (a 1e-7 change of re / im moves the phase of a 1e-4 bin by 1e-3), and what the generator consumes
    is conditioned like magnitude * phase error."""
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    d[11:] = np.minimum(d[11:], np.abs(2 * np.pi - d[11:])) * np.minimum(1.0, 50.0 * np.abs(ref[:11].astype(np.float64)))
    return d


def _inputs(token_counts, seed0=10):
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    ids = [R.synthetic_inputs(1, n, seed=seed0 + i)[0] for i, n in enumerate(token_counts)]
    voices = W.synthetic_voices(4)
    styles = [voices[i % 4, n, 0] for i, n in enumerate(token_counts)]
    return ids, styles


@pytest.mark.parametrize("mode", [pytest.param(0, id="f32"), pytest.param(1, id="f16x3"), pytest.param(6, id="f16f8")])
def test_ragged_batch_matches_oracle(hip_model, oracle, mode):
    from kokorox_amd import hip_koko as hk
    default_mode = hip_model.get_conv_mode()
    hip_model.set_conv_mode(mode)
    try:
        _ragged_batch_vs_oracle(hip_model, oracle)
    finally:
        hip_model.set_conv_mode(default_mode)


def _ragged_batch_vs_oracle(hip_model, oracle):
    from kokorox_amd import hip_koko as hk
    counts = [21, 9, 14]
    ids, styles = _inputs(counts)
    outs = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=2, flags=hk.KX_FLAG_TAPS)
    for b in range(len(counts)):
        taps = {}
        audio, dur = oracle.forward(ids[b], styles[b], 1.0, seed=2, utt=b, taps=taps)
        assert len(outs[b]) == 600 * int(dur.sum()), "predicted durations differ"
        for name in FRONT:
            ref = taps[name].numpy()
            got = hip_model.tap(name, b)
            assert got.shape == ref.shape, name
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name
        f0 = hip_model.tap("pred.F0", b)
        ref_f0 = taps["pred.F0"].numpy()
        assert np.abs(f0 - ref_f0).max() <= 5e-5 * np.abs(ref_f0).max()
        # edge 1 pinned: the GPU's F0 / N curves -> harmonic source (bit-faithful phase) and STFT
        n_c = hip_model.tap("pred.N", b)[0]
        taps2 = {}
        oracle.forward(ids[b], styles[b], 1.0, seed=2, utt=b, taps=taps2, f0_override=f0[0], n_override=n_c)
        assert np.abs(hip_model.tap("gen.har_source", b) - taps2["gen.har_source"].numpy()).max() < 2e-6
        har = hip_model.tap("gen.har", b)
        assert _wrapped_diff(har, taps2["gen.har"].numpy()).max() < 2e-3
        # edge 2 pinned: the GPU's STFT (atan2 branch cut) -> everything downstream incl. the waveform
        taps3 = {}
        audio3, _ = oracle.forward(ids[b], styles[b], 1.0, seed=2, utt=b, taps=taps3, f0_override=f0[0],
                                   n_override=n_c, har_override=har)
        worst = 0.0
        for name in BACK:
            ref = taps3[name].numpy()
            got = hip_model.tap(name, b)
            assert got.shape == ref.shape, name
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name
            worst = max(worst, float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max())))
        wave = float(np.abs(outs[b] - audio3.numpy()).max())
        print(f"conv mode {hip_model.get_conv_mode()}, utterance {b}: back-half taps within {worst:.2e} (relative to max(1, |tap|)), "
              f"waveform max|d| {wave:.2e}")  # (pytest -s: the figures DESIGN.md section 4 quotes)
        assert wave < TOL_WAVE
        assert np.abs(hip_model.tap("audio", b)[0] - outs[b]).max() == 0.0


def test_golden_vectors(hip_model, golden):
    """Committed fixtures (tests/golden, tools/make_golden.py): no oracle run needed for the front half."""
    from kokorox_amd import hip_koko as hk
    for name, g in golden.items():
        out = hip_model.infer([list(g["ids"])], [list(g["style"])], float(g["speed"]), seed=int(g["seed"]),
                              flags=hk.KX_FLAG_TAPS)
        assert out.shape[0] == 600 * int(g["pred_dur"].sum()), name
        for k in ("d_en", "dur.lstm", "text_enc.out", "pred.N"):
            ref = g["tap:" + k]
            assert np.abs(hip_model.tap(k, 0) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (name, k)
        ref = g["tap:pred.F0"]
        assert np.abs(hip_model.tap("pred.F0", 0) - ref).max() <= 5e-5 * np.abs(ref).max()


def test_batch_of_one_equals_member_of_batch(hip_model):
    """The reference is batch-1 (koko.rs:1175); batching must not change any utterance: bit-exact."""
    counts = [17, 30, 12, 30]
    ids, styles = _inputs(counts, seed0=40)
    hip_model.set_utterance_base(0)
    batch = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=9)
    for b in (0, 2, 3):
        hip_model.set_utterance_base(b)
        alone = hip_model.infer([list(ids[b])], [list(styles[b])], 1.0, seed=9)
        np.testing.assert_array_equal(alone, batch[b])
    hip_model.set_utterance_base(0)


def test_lanes_do_not_change_a_bit(hip_model):
    """The back half issues its independent chains (noise path, the three resblocks of a stage) on streams of their
    own (kx_set_lanes); only the last conv of a chain touches the shared running sum, in a fixed order, so one stream
    and four must give the same bits -- on a batch large enough that the chains really overlap."""
    counts = [40, 23, 40, 31, 17, 40]
    ids, styles = _inputs(counts, seed0=300)
    hip_model.set_lanes(1)
    try:
        one = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=5)
        hip_model.set_lanes(4)
        for _ in range(2):
            four = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=5)
            for a, b in zip(one, four):
                np.testing.assert_array_equal(a, b)
    finally:
        hip_model.set_lanes(0)


def test_lstm_handoff_timeout_is_survived_without_changing_a_bit(blob_path):
    """A part of a resident-weights LSTM recurrence whose partner never shows up raises the sticky device error word and leaves:
    what that call computed is garbage.  Round 5: the model switches to the streaming recurrence -- which gives the SAME BITS --
    and the host entry points run the call once more, so the caller gets the undisturbed result, not an error; kx_model_status
    reports the time-out, the fall-back and the re-run, and the resident forms return after 64 clean forwards.  On the
    device-pointer path a time-out of a token-axis recurrence is re-run the same way (seen at the forward's one host wait); one of
    the frame-axis recurrence shows only at kx_sync, which must fail with KX_ERR_DEVICE (never hand out garbage)."""
    from kokorox_amd import hip_koko as hk
    import torch
    ids, styles = _inputs([12], seed0=310)
    m = hk.HipKoko.new(blob_path)
    names = ("text_enc.out", "dur.lstm", "pred.F0", "pred.N", "dec.decode.3")
    lib = hk.load_test_library()
    try:
        good = m.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4, flags=hk.KX_FLAG_TAPS)
        good_taps = {n: m.tap(n, 0).copy() for n in names}
        assert m.status() == [0, 0, 0, 0]
        assert lib.kx_test_lstm_fault(6) != 0, "the fault hook must not arm outside a test process (KX_TEST_HOOKS unset)"
        os.environ["KX_TEST_HOOKS"] = "1"
        try:
            # (1) host path, the sixth recurrence of the forward = predictor.shared over the frame axis, after the mid-way check
            assert lib.kx_test_lstm_fault(6) == 0
            out = m.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4)
            np.testing.assert_array_equal(out, good)
            assert m.status() == [1, 1, 63, 1], m.status()
            lib.kx_test_lstm_fault(0)
            # (2) the streaming recurrence from here on: every tap and the waveform bit for bit
            again = m.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4, flags=hk.KX_FLAG_TAPS)
            np.testing.assert_array_equal(again, good)
            for n in names:
                np.testing.assert_array_equal(m.tap(n, 0), good_taps[n])
            # (3) the resident forms return after 64 clean forwards
            for _ in range(62):
                m.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4)
            assert m.status() == [0, 1, 0, 1], m.status()
            np.testing.assert_array_equal(m.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4), good)
            # (4) device-pointer path
            dev = torch.device("cuda:0")
            T = len(ids[0])
            d_ids = torch.tensor([list(ids[0])], dtype=torch.int64, device=dev)
            d_st = torch.tensor(np.asarray([styles[0]], dtype=np.float32), device=dev)
            d_audio = torch.zeros((1, good.shape[0]), dtype=torch.float32, device=dev)
            d_fr = torch.zeros(1, dtype=torch.int32, device=dev)
            lens = np.array([T], dtype=np.int32)
            torch.cuda.synchronize()
            assert lib.kx_test_lstm_fault(2) == 0  # a token-axis recurrence: re-run inside kx_infer_device
            m.infer_device(d_ids.data_ptr(), T, lens, d_st.data_ptr(), np.array([1.0], dtype=np.float32), d_audio.data_ptr(), good.shape[0],
                           d_fr.data_ptr(), seed=4)
            lib.kx_test_lstm_fault(0)
            m.sync()
            np.testing.assert_array_equal(d_audio[0].cpu().numpy(), good)
            st = m.status()
            assert st[0] == 1 and st[1] == 2 and st[3] == 2, st
            # ... and the frame-axis one: the call cannot be repeated from kx_sync, which must fail
            for _ in range(64):
                m.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4)
            assert m.status()[0] == 0
            assert lib.kx_test_lstm_fault(6) == 0
            m.infer_device(d_ids.data_ptr(), T, lens, d_st.data_ptr(), np.array([1.0], dtype=np.float32), d_audio.data_ptr(), good.shape[0],
                           d_fr.data_ptr(), seed=4)
            with pytest.raises(hk.KokoroxHipError) as ei:
                m.sync()
            assert "LSTM" in str(ei.value)
            lib.kx_test_lstm_fault(0)
            np.testing.assert_array_equal(m.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4), good)
        finally:
            lib.kx_test_lstm_fault(0)
            del os.environ["KX_TEST_HOOKS"]
    finally:
        m.close()


@pytest.mark.parametrize("mode,name,tol", [(4, "f16", REDUCED_TOL), (5, "bf16", REDUCED_TOL_BF16)])
def test_reduced_precision_mode_error_is_bounded(hip_model, oracle, mode, name, tol):
    """The opt-in reduced-precision modes (kx_set_conv_mode(4) / KOKOROX_CONV=f16 and, round 5, kx_set_conv_mode(5) /
    KOKOROX_CONV=bf16 -- the dtype BASELINE configs[2] names: one 16-bit MFMA per product in the decoder and generator convs; the
    reference's counterpart are its model_fp16 / quantised variants, hf_cache.rs:135-144).  They are NOT inside the 1e-4 parity
    band and never the default; what they must keep: durations and the F0 / N curves exactly as in the default mode (the front
    half stays f32-class), and a waveform within the bound measured on this workload (profiles/r03_reduced_precision.txt for
    f16; bf16 carries 8 significant bits instead of 11: 8x the bound), under the usual protocol."""
    from kokorox_amd import hip_koko as hk
    ids, styles = _inputs([24], seed0=500)
    base = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=2, flags=hk.KX_FLAG_TAPS)
    f0_base = hip_model.tap("pred.F0", 0)
    default_mode = hip_model.get_conv_mode()
    hip_model.set_conv_mode(mode)
    try:
        assert hip_model.get_conv_mode() == mode
        out = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=2, flags=hk.KX_FLAG_TAPS)
        f0, n_c, har = hip_model.tap("pred.F0", 0), hip_model.tap("pred.N", 0)[0], hip_model.tap("gen.har", 0)
        np.testing.assert_array_equal(f0, f0_base)  # nothing upstream of the curves is reduced
        assert out.shape == base.shape
        audio, dur = oracle.forward(ids[0], styles[0], 1.0, seed=2, utt=0, f0_override=f0[0], n_override=n_c, har_override=har)
        err = float(np.abs(out - audio.numpy()).max())
        scale = float(np.abs(audio.numpy()).max())
        print(f"reduced precision {name}: max|d| vs the oracle {err:.3e} (waveform scale {scale:.3f})")
        assert 1e-6 < err < tol * max(scale, 1.0), (err, scale)  # really reduced, and bounded
        if mode == 5:
            assert err > 2e-4, err  # (coarser than f16's measured 3 - 3.5e-4 would be expected; far outside the 1e-4 band in any case)
    finally:
        hip_model.set_conv_mode(default_mode)
    again = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=2)
    np.testing.assert_array_equal(again, base)


def test_f16f8_mode_against_the_f16x3_mode_over_a_spread_of_inputs(hip_model):
    """The default mode differs from the f16x3 mode in the generator's 7- / 11-tap convs (8-bit cross terms) and in the snake's sin^2
    (hardware cosine) only: durations, F0 / N curves and harmonic source are the same bits, and the waveform stays within 3e-5 --
    a third of the parity band -- over sixteen utterances of 5 .. 200 phonemes, four voices, three speeds."""
    from kokorox_amd import hip_koko as hk
    counts = [5, 9, 17, 33, 48, 64, 80, 96, 112, 128, 150, 200, 23, 71, 128, 37]
    ids, styles = _inputs(counts, seed0=900)
    speeds = [1.0, 0.8, 1.3]
    default_mode = hip_model.get_conv_mode()
    worst = 0.0
    try:
        for i, n in enumerate(counts):
            outs, f0s = {}, {}
            for mode in (1, 6):
                hip_model.set_conv_mode(mode)
                outs[mode] = hip_model.infer([list(ids[i])], [list(styles[i])], speeds[i % 3], seed=40 + i, flags=hk.KX_FLAG_TAPS)
                f0s[mode] = (hip_model.tap("pred.F0", 0), hip_model.tap("gen.har_source", 0))
            assert outs[1].shape == outs[6].shape, n
            np.testing.assert_array_equal(f0s[1][0], f0s[6][0])
            np.testing.assert_array_equal(f0s[1][1], f0s[6][1])
            assert np.isfinite(outs[6]).all()
            worst = max(worst, float(np.abs(outs[1] - outs[6]).max()))
    finally:
        hip_model.set_conv_mode(default_mode)
    print(f"f16f8 against f16x3 over {len(counts)} utterances: waveform max|d| {worst:.2e}")
    assert 0 < worst < 3e-5


def test_determinism_and_seed(hip_model):
    ids, styles = _inputs([25], seed0=70)
    a = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=1)
    b = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=1)
    c = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=2)
    np.testing.assert_array_equal(a, b)
    assert a.shape == c.shape and np.abs(a - c).max() > 1e-4


def test_speed_and_per_utterance_speeds(hip_model):
    ids, styles = _inputs([20, 20], seed0=80)
    ids[1], styles[1] = ids[0], styles[0]
    outs = hip_model.infer_batch([list(x) for x in ids], styles, [1.0, 2.0], seed=3)
    assert len(outs[1]) < len(outs[0]) and len(outs[1]) % 600 == 0
    fast = hip_model.infer([list(ids[0])], [list(styles[0])], 2.0, seed=3)
    assert len(fast) == len(outs[1])


def test_pinned_durations_and_full_size_properties(hip_model):
    """BASELINE workload shape (T=130, F=422) at a bounded batch: lengths, finiteness, range, repeatability."""
    from oracle import kokoro_ref as R
    from kokorox_amd import weights as W
    B = 4
    ids = R.synthetic_inputs(B, 128, seed=0)
    voices = W.synthetic_voices(4)
    styles = [voices[b, 128, 0] for b in range(B)]
    hip_model.set_pinned_durations([3, 3, 3, 4])
    try:
        outs = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=2)
        again = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=2)
    finally:
        hip_model.set_pinned_durations(None)
    F = int(R.pinned_durations(130).sum())
    assert F == 422
    for a, b in zip(outs, again):
        assert a.shape[0] == 600 * F
        assert np.isfinite(a).all() and np.abs(a).max() < 10.0 and a.std() > 1e-3
        np.testing.assert_array_equal(a, b)


def _teacher_forced_oracle_check(hip_model, oracle, ids, style, got_wave, utt, pinned):
    """The parity protocol of _ragged_batch_vs_oracle on ONE utterance whose taps are in the model (B = 1 call)."""
    taps = {}
    _, dur = oracle.forward(ids, style, 1.0, seed=2, utt=utt, taps=taps, pinned_dur=pinned)
    assert got_wave.shape[0] == 600 * int(dur.sum())
    for name in FRONT:
        ref = taps[name].numpy()
        got = hip_model.tap(name, 0)
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name
    f0 = hip_model.tap("pred.F0", 0)
    ref_f0 = taps["pred.F0"].numpy()
    assert np.abs(f0 - ref_f0).max() <= 5e-5 * np.abs(ref_f0).max()
    n_c = hip_model.tap("pred.N", 0)[0]
    taps2 = {}
    oracle.forward(ids, style, 1.0, seed=2, utt=utt, taps=taps2, pinned_dur=pinned, f0_override=f0[0], n_override=n_c)
    assert np.abs(hip_model.tap("gen.har_source", 0) - taps2["gen.har_source"].numpy()).max() < 2e-6
    har = hip_model.tap("gen.har", 0)
    assert _wrapped_diff(har, taps2["gen.har"].numpy()).max() < 2e-3
    taps3 = {}
    audio3, _ = oracle.forward(ids, style, 1.0, seed=2, utt=utt, taps=taps3, pinned_dur=pinned, f0_override=f0[0],
                               n_override=n_c, har_override=har)
    for name in BACK:
        ref = taps3[name].numpy()
        got = hip_model.tap(name, 0)
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name
    assert np.abs(got_wave - audio3.numpy()).max() < TOL_WAVE


def test_baseline_workload_at_size_batch64_and_batch1(hip_model, oracle):
    """BASELINE configs[1] and configs[2] at full size (T = 130, durations pinned 3,3,3,4 -> F = 422):
      * B = 64 in one call: frames, finiteness;
      * every one of the 64 equals, bit for bit, its own B = 1 call (that call runs the small-grid tile path at
        full length: one 128-phoneme utterance on one MI355X);
      * utterances 0, 31 and 63 are compared with the CPU oracle under the teacher-forcing protocol (every tap and
        the waveform, |d| < 1e-4)."""
    from oracle import kokoro_ref as R
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    B = 64
    ids = R.synthetic_inputs(B, 128, seed=0)
    voices = W.synthetic_voices(4)
    styles = [voices[b % 4, 128, 0] for b in range(B)]
    pinned = R.pinned_durations(130)
    hip_model.set_pinned_durations([3, 3, 3, 4])
    try:
        hip_model.set_utterance_base(0)
        outs = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=2)
        assert len(outs) == B
        for b in range(B):
            assert outs[b].shape[0] == 600 * 422 and np.isfinite(outs[b]).all()
        for b in range(B):
            hip_model.set_utterance_base(b)  # noise key (seed, b), as inside the batch
            checked = b in (0, 31, 63)
            one = hip_model.infer([list(ids[b])], [styles[b]], 1.0, seed=2, flags=hk.KX_FLAG_TAPS if checked else 0)
            np.testing.assert_array_equal(one, outs[b], err_msg=f"utterance {b}: batch of 64 != batch of 1")
            if checked:
                _teacher_forced_oracle_check(hip_model, oracle, ids[b], styles[b], one, b, pinned)
    finally:
        hip_model.set_utterance_base(0)
        hip_model.set_pinned_durations(None)


def test_ragged_batch64_at_size_equals_batch1(hip_model, oracle):
    """The same 64 utterances with their PREDICTED durations (the bench's free-running step: 390 .. 892 frames, 38 507 in all):
    the conv kernels then take the flat list of live tiles over a really ragged batch at full size.  The shortest, the longest
    and two others must equal their own B = 1 call (which takes the dense single-utterance grid) bit for bit, the frame counts
    must be the oracle's durations, and one of them is compared with the oracle, every tap and the waveform."""
    from oracle import kokoro_ref as R
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    B = 64
    ids = R.synthetic_inputs(B, 128, seed=0)
    voices = W.synthetic_voices(4)
    styles = [voices[b % 4, 128, 0] for b in range(B)]
    hip_model.set_pinned_durations(None)
    hip_model.set_utterance_base(0)
    try:
        outs = hip_model.infer_batch([list(x) for x in ids], styles, [1.0], seed=2)
        frames = np.array([o.shape[0] // 600 for o in outs])
        assert frames.min() < frames.max() and all(np.isfinite(o).all() for o in outs)
        picks = sorted({int(frames.argmin()), int(frames.argmax()), 17, 40})
        for b in picks:
            hip_model.set_utterance_base(b)
            checked = b == picks[0]
            one = hip_model.infer([list(ids[b])], [styles[b]], 1.0, seed=2, flags=hk.KX_FLAG_TAPS if checked else 0)
            np.testing.assert_array_equal(one, outs[b], err_msg=f"utterance {b} ({frames[b]} frames): ragged batch of 64 != batch of 1")
            if checked:
                _teacher_forced_oracle_check(hip_model, oracle, ids[b], styles[b], one, b, None)
    finally:
        hip_model.set_utterance_base(0)


def test_warmup_sizes_the_arenas_ahead_of_the_first_request(blob_path):
    """kx_warmup (for servers): after it, a batch inside the warmed shape must not grow any arena (a regrowth under load is a
    stream sync + hipFree + hipMalloc of gigabytes: the 1.3 s outlier of profiles/r04_serve_models_per_gpu.txt), a pinned
    pattern set before must survive it, and results must not depend on it."""
    from kokorox_amd import hip_koko as hk
    ids, styles = _inputs([40, 25, 33, 12, 40, 9, 31, 38], seed0=123)
    m = hk.HipKoko.new(blob_path)
    try:
        assert m.arena_bytes() == [0, 0, 0]
        m.warmup(8, 42, 8)
        cap = m.arena_bytes()
        assert all(c > 0 for c in cap)
        out = m.infer_batch([list(x) for x in ids], [list(s) for s in styles], [1.0], seed=3)
        assert m.arena_bytes() == cap, "a batch inside the warmed shape grew an arena"
        m2 = hk.HipKoko.new(blob_path)
        try:
            ref = m2.infer_batch([list(x) for x in ids], [list(s) for s in styles], [1.0], seed=3)
        finally:
            m2.close()
        for a, b in zip(out, ref):
            np.testing.assert_array_equal(a, b)
        m.set_pinned_durations([2, 3])
        m.warmup(2, 20, 5)
        one = m.infer([list(ids[3])], [list(styles[3])], 1.0, seed=3)
        assert one.shape[0] == 600 * sum([2, 3][i % 2] for i in range(len(ids[3])))
        with pytest.raises(hk.KokoroxHipError, match="warmup"):
            m.warmup(0, 20, 5)
    finally:
        m.close()


def test_device_entry_point_rejects_bad_ids(hip_model):
    """kx_infer_device takes ids from device memory: the embedding kernels range-check them (clamped gather, sticky
    error word) and the call returns KX_ERR_INVALID instead of reading out of bounds."""
    import torch
    from kokorox_amd import hip_koko as hk
    ids = torch.tensor([[0, 5, 500, 7, 0], [0, 9, 9, 9, 0]], dtype=torch.int64, device="cuda")
    styles = torch.zeros((2, 256), dtype=torch.float32, device="cuda")
    audio = torch.zeros((2, 600 * 5 * 50), dtype=torch.float32, device="cuda")
    frames = torch.zeros(2, dtype=torch.int32, device="cuda")
    lens = np.array([5, 5], dtype=np.int32)
    with pytest.raises(hk.KokoroxHipError, match="token id outside 0..177.*utterance 0, position 2") as ei:
        hip_model.infer_device(ids.data_ptr(), 5, lens, styles.data_ptr(), np.array([1.0], np.float32),
                               audio.data_ptr(), audio.shape[1], frames.data_ptr())
    assert ei.value.code == 1
    ids[0, 2] = -3
    with pytest.raises(hk.KokoroxHipError, match="token id"):
        hip_model.infer_device(ids.data_ptr(), 5, lens, styles.data_ptr(), np.array([1.0], np.float32),
                               audio.data_ptr(), audio.shape[1], frames.data_ptr())
    ids[0, 2] = 6  # the same buffers work once the id is valid
    hip_model.infer_device(ids.data_ptr(), 5, lens, styles.data_ptr(), np.array([1.0], np.float32),
                           audio.data_ptr(), audio.shape[1], frames.data_ptr())
    hip_model.sync()
    assert int(frames.min()) >= 5 and torch.isfinite(audio).all()


def test_last_error_is_a_per_thread_copy(hip_model):
    """kx_last_error is the CALLING thread's own last failure on the model: with three threads on one model -- two failing
    with different messages, one succeeding all the time -- nobody's message is blanked by another thread's success or
    replaced by another thread's failure."""
    import threading
    from kokorox_amd import hip_koko as hk
    ids, styles = _inputs([6], seed0=900)
    style = [0.0] * 256
    seen = {"id": [], "speed": []}
    ok = []

    def bad_id():
        for _ in range(15):
            try:
                hip_model.infer([[0, 500, 0]], [style], 1.0)
            except hk.KokoroxHipError as e:
                seen["id"].append(str(e))

    def bad_speed():
        for _ in range(15):
            try:
                hip_model.infer([[0, 5, 0]], [style], -1.0)
            except hk.KokoroxHipError as e:
                seen["speed"].append(str(e))

    def good():
        for _ in range(15):
            ok.append(hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=1).shape[0])

    th = [threading.Thread(target=f) for f in (bad_id, bad_speed, good)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert len(seen["id"]) == 15 and all("token id outside" in m for m in seen["id"]), seen["id"][:3]
    assert len(seen["speed"]) == 15 and all("speed must be" in m for m in seen["speed"]), seen["speed"][:3]
    assert len(ok) == 15 and len(set(ok)) == 1
    assert hip_model.last_error() == ""  # (this thread has not failed)


def test_activation_prescale_and_diagnostics_cover_every_conv(hip_model):
    """Real-weights kit.  A power-of-two activation pre-scale only moves the f16 subnormal quantum of the low halves
    (2^-25 absolute): results change at the 1e-7..1e-6 level (the recurrences amplify it a little), never more (the waveform itself is not compared: it sits behind
    the F0 -> phase edge).  The diagnostics see every conv launch with sane magnitudes on the synthetic
    weights."""
    from kokorox_amd import hip_koko as hk
    ids, styles = _inputs([19], seed0=70)
    taps = ["bert.out", "dur.lstm", "pred.shared", "text_enc.out", "dec.encode", "dec.decode.3"]
    hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=5, flags=hk.KX_FLAG_TAPS)
    ref = {t: hip_model.tap(t, 0) for t in taps + ["pred.F0"]}
    layers = ["decoder.decode.1.conv2", "bert.ffn", "predictor.lstm.ih", "decoder.encode.conv1", "text_encoder.cnn.1",
              "decoder.generator.resblocks.3.convs1.0", "decoder.generator.ups.1"]
    try:
        for e, name in zip((3, -2, 5, 1, 4, 2, 3), layers):
            hip_model.set_act_prescale(name, e)
        out = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=5, flags=hk.KX_FLAG_TAPS)
        got = {t: hip_model.tap(t, 0) for t in taps + ["pred.F0"]}
    finally:
        for name in layers:
            hip_model.set_act_prescale(name, 0)
    assert np.isfinite(out).all()
    for t in taps:
        assert np.abs(got[t] - ref[t]).max() <= 1e-5 * max(1.0, np.abs(ref[t]).max()), t
    assert np.abs(got["pred.F0"] - ref["pred.F0"]).max() <= 2e-5 * np.abs(ref["pred.F0"]).max()
    with pytest.raises(Exception, match="no such conv layer"):
        hip_model.set_act_prescale("decoder.nonexistent", 1)
    hip_model.diag_enable(True)
    try:
        hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=5)
        recs = hip_model.diag_records()
    finally:
        hip_model.diag_enable(False)
    names = {r[0] for r in recs}
    assert len(recs) >= 140 and "decoder.generator.conv_post" in names and "bert.qkv" in names
    for name, rows, cin, k, sh, amax, rms, cnt in recs:
        assert np.isfinite(amax) and np.isfinite(rms) and cnt > 0, name
        assert amax < 6e4 and (rms > 1e-3 or "F0_conv" in name or "N_conv" in name), (name, amax, rms)


def test_max_length_utterance(hip_model):
    """510 tokens + 2 pads is the model limit (voice table rows, hf_cache.rs:302-309)."""
    from oracle import kokoro_ref as R
    from kokorox_amd import weights as W
    ids = R.synthetic_inputs(1, 510, seed=5)[0]
    style = W.synthetic_voices(1)[0, 510, 0]
    hip_model.set_pinned_durations([1])
    try:
        out = hip_model.infer([list(ids)], [list(style)], 1.0, seed=0)
    finally:
        hip_model.set_pinned_durations(None)
    assert out.shape[0] == 600 * 512 and np.isfinite(out).all()


def test_error_behaviour(hip_model, blob_path, tmp_path):
    """Errors are statuses with messages, never aborts."""
    from kokorox_amd import hip_koko as hk
    style = [0.0] * 256
    with pytest.raises(ValueError):
        hip_model.infer([], [], 1.0)                         # ort_koko.rs:56 would panic here
    with pytest.raises(ValueError):
        hip_model.infer([[]], [style], 1.0)
    with pytest.raises(hk.KokoroxHipError, match="token id"):
        hip_model.infer([[0, 500, 0]], [style], 1.0)
    with pytest.raises(hk.KokoroxHipError, match="speed"):
        hip_model.infer([[0, 5, 0]], [style], 0.0)
    with pytest.raises(hk.KokoroxHipError, match="1..512"):
        hip_model.infer([[0] * 513], [style], 1.0)
    with pytest.raises(RuntimeError, match="cannot open"):
        hk.HipKoko.new(str(tmp_path / "missing.kxw"))
    bad = tmp_path / "bad.kxw"
    bad.write_bytes(b"NOTABLOB" + b"\0" * 100)
    with pytest.raises(RuntimeError, match="bad magic"):
        hk.HipKoko.new(str(bad))
    # the model is still usable after errors
    assert hip_model.infer([[0, 5, 6, 0]], [style], 1.0).shape[0] % 600 == 0


def test_device_blob_constructor_matches_file_constructor(hip_model, blob_path):
    import torch
    from kokorox_amd import hip_koko as hk
    buf = torch.from_numpy(np.fromfile(blob_path, dtype=np.uint8)).cuda()
    m2 = hk.HipKoko.from_device_blob(buf.data_ptr(), buf.numel(), 0)
    del buf
    ids, styles = _inputs([13], seed0=90)
    a = hip_model.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4)
    b = m2.infer([list(ids[0])], [list(styles[0])], 1.0, seed=4)
    m2.close()
    np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_chunks_of_a_long_text_batched_equal_chunk_by_chunk(hip_model):
    """SURVEY §8 row a8 (koko.rs:947-1191): the reference runs the chunks of a long text one after the other; the
    host mirror batches them.  Chunk b of the batch draws its noise from key (seed, b), so it must equal, bit for
    bit, a batch-of-one call whose utterance base is b."""
    from kokorox_amd import voices as V
    from kokorox_amd import weights as W

    table = {"af_sky": W.synthetic_voices(2)[0], "af_nicole": W.synthetic_voices(2)[1]}
    rng = np.random.default_rng(77)
    chunks = [rng.integers(1, 178, size=n).tolist() for n in (41, 7, 120)]
    style = "af_sky.4+af_nicole.5"
    whole = V.tts_chunks(hip_model, table, style, chunks, speed=1.0, initial_silence=1, seed=11)
    parts = []
    for b, ch in enumerate(chunks):
        t = [30] + ch
        hip_model.set_utterance_base(b)
        parts.append(hip_model.infer([[0] + t + [0]], V.mix_styles(table, style, len(t)), 1.0, seed=11))
    hip_model.set_utterance_base(0)
    ref = np.concatenate(parts)
    assert whole.shape == ref.shape and whole.shape[0] > 24000
    assert np.array_equal(whole, ref)
    # one chunk against the CPU oracle as well (not only the HIP path against itself): chunk 1 of the batch = utterance
    # index 1 of the noise stream, its style row mixed for its own token count (koko.rs:1165)
    from kokorox_amd import hip_koko as hk
    from oracle import kokoro_ref as R
    o = R.KokoroOracle(W.ensure_synthetic_blob())
    t = [30] + chunks[1]
    ids = np.array([0] + t + [0], dtype=np.int64)
    row = np.asarray(V.mix_styles(table, style, len(t))[0], dtype=np.float32)
    hip_model.set_utterance_base(1)
    got = hip_model.infer([list(ids)], [list(row)], 1.0, seed=11, flags=hk.KX_FLAG_TAPS)
    hip_model.set_utterance_base(0)
    np.testing.assert_array_equal(got, parts[1])
    f0, n_c, har = hip_model.tap("pred.F0", 0), hip_model.tap("pred.N", 0)[0], hip_model.tap("gen.har", 0)
    audio, dur = o.forward(ids, row, 1.0, seed=11, utt=1, f0_override=f0[0], n_override=n_c, har_override=har)
    assert got.shape[0] == 600 * int(dur.sum())
    assert np.abs(got - audio.numpy()).max() < TOL_WAVE


def test_kx_init_on_a_real_device():
    """`init_ort` (kokorox/src/onn/mod.rs:19-49) loads the runtime and reports failure as a string; its counterpart checks that the
    device exists and is a gfx950: a valid id succeeds, ids outside the visible devices fail with a message, nothing panics."""
    import ctypes as C
    from kokorox_amd import hip_koko as hk
    lib = hk.load_library()
    err = C.create_string_buffer(256)
    assert lib.kx_init(0, err, len(err)) == 0, err.value
    assert lib.kx_init(99, err, len(err)) == 1 and b"device id out of range" in err.value, err.value
    assert lib.kx_init(-1, err, len(err)) == 1 and b"device id out of range" in err.value, err.value
    assert lib.kx_init(0, None, 0) == 0  # (no message buffer is fine)


def test_dispatcher_warm_up_sizes_the_arenas_before_the_first_request(blob_path):
    """kx_dispatcher_create_warm: one discarded forward of the largest expected shape per model, so the first real batch finds its
    arenas (round 4's soak saw a 1.3 s outlier where an arena grew under a request)."""
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    m = hk.HipKoko.new(blob_path)
    try:
        assert m.arena_bytes() == [0, 0, 0]
        d = hk.Dispatcher([m], max_batch=8, max_wait_us=500, warm=(64, 6))
        try:
            before = m.arena_bytes()
            assert min(before) > 0
            ids = R.synthetic_inputs(1, 40, seed=8)[0]
            style_row = W.synthetic_voices(1)[0, 40, 0]
            out = d.submit(list(ids), list(style_row), 1.0, seed=2)
            assert out.shape[0] > 0 and np.isfinite(out).all()
            assert m.arena_bytes() == before, "an arena grew under the first request"
        finally:
            d.close()
    finally:
        m.close()
