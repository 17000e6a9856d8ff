"""GPU: SURVEY 8f ranks 2 and 3 — style rows looked up / mixed on the device from a resident voice table,
and the reference's output forms packed on the device.  Both must equal their host mirrors bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(hip_model):
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    tab = W.synthetic_voices(5)
    names = ["af_sky", "af_nicole", "am_adam", "bf_emma", "jf_alpha"]
    styles = {n: tab[i] for i, n in enumerate(names)}
    hip_model.set_voice_table(tab)
    hip_model.set_utterance_base(0)
    toks = [list(R.synthetic_inputs(1, k, seed=300 + k)[0]) for k in (12, 20, 9)]
    return names, styles, toks


def test_device_style_mix_equals_host_mixer(hip_model):
    from kokorox_amd import voices as V
    names, styles, toks = _setup(hip_model)
    # utterance 0: "af_sky.4+af_nicole.5"; 1: "bf_emma.7+junk+am_adam.3.5" (middle part skipped); 2: "jf_alpha.10"
    vid = [[0, 1, -1], [3, -1, 2], [4, -1, -1]]
    wts = [[4, 5, 0], [7, 0, 3.5], [10, 0, 0]]
    host_styles = [V.mix_styles(styles, "af_sky.4+af_nicole.5", 12)[0],
                   V.mix_styles(styles, "bf_emma.7+junk+am_adam.3.5", 20)[0],
                   V.mix_styles(styles, "jf_alpha.10+zzz", 9)[0]]
    want = hip_model.infer_batch(toks, host_styles, [1.0], seed=5)
    got = hip_model.infer_voices(toks, vid, wts, [1.0], seed=5)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    # single-voice form: a plain copy of row tokens_len (no x1.0)
    one = hip_model.infer_voices(toks[:1], [[1]], [[0.0]], [1.0], seed=5)[0]
    ref = hip_model.infer([toks[0]], V.mix_styles(styles, "af_nicole", 12), 1.0, seed=5)
    np.testing.assert_array_equal(one, ref)
    from kokorox_amd import hip_koko as hk
    with pytest.raises(hk.KokoroxHipError, match="voice id"):
        hip_model.infer_voices(toks[:1], [[99]], [[1.0]])


def test_packed_outputs_match_reference_conversions(hip_model):
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import voices as V
    names, styles, toks = _setup(hip_model)
    st = [V.mix_styles(styles, "af_sky", len(t) - 2)[0] for t in toks]
    mono = hip_model.infer_batch(toks, st, [1.0], seed=8)
    stereo = hip_model.infer_packed(toks, st, [1.0], seed=8, fmt=hk.PACK_F32_STEREO)
    pcm = hip_model.infer_packed(toks, st, [1.0], seed=8, fmt=hk.PACK_PCM16_MONO)
    for m, s2, p in zip(mono, stereo, pcm):
        assert s2.shape == (m.shape[0], 2) and p.dtype == np.int16
        np.testing.assert_array_equal(s2[:, 0], m)          # koko.rs:1239-1246: sample written twice
        np.testing.assert_array_equal(s2[:, 1], m)
        want = np.trunc(np.clip(m, -1.0, 1.0) * np.float32(32767.0)).astype(np.int16)  # websocket lib.rs:701-704
        np.testing.assert_array_equal(p, want)
    # voices + packing together
    both = hip_model.infer_voices(toks[:1], [[0]], [[0.0]], [1.0], seed=8, fmt=hk.PACK_PCM16_MONO)[0]
    np.testing.assert_array_equal(both, pcm[0])
