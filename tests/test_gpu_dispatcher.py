"""GPU: the request dispatcher.  Concurrent clients are coalesced into batches, and each
request's waveform is bit-identical to the same request run alone through kx_infer."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_requests_are_batched_and_batch_invariant(hip_model):
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    n = 12
    voices = W.synthetic_voices(4)
    reqs = []
    for i in range(n):
        k = 8 + 3 * i
        reqs.append((R.synthetic_inputs(1, k, seed=200 + i)[0], voices[i % 4, k, 0], 1.0 + 0.1 * (i % 3), 1000 + i))
    d = hk.Dispatcher([hip_model], max_batch=8, max_wait_us=200000)
    out = [None] * n
    errs = []

    def client(i):
        try:
            ids, style, speed, seed = reqs[i]
            out[i] = d.submit(ids, style, speed, seed)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=client, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    st = d.stats()
    with pytest.raises(hk.KokoroxHipError):
        d.submit([0, 999, 0], reqs[0][1], 1.0, 0)
    d.close()
    assert not errs, errs
    assert st["requests"] == n and st["batches"] < n and st["max_batch"] > 1
    hip_model.set_utterance_base(0)
    for i, (ids, style, speed, seed) in enumerate(reqs):
        alone = hip_model.infer([list(ids)], [list(style)], speed, seed=seed)
        np.testing.assert_array_equal(out[i], alone)


def test_replicas_share_one_file_read_and_serve_one_dispatcher(blob_path, hip_model):
    """kx_create_replicas: two models from one read of the weight file (both on device 0 here), bit-identical to a
    model made by kx_create, and a dispatcher over BOTH of them (n_models > 1) keeps the batch-invariance."""
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    ms = hk.HipKoko.replicas(blob_path, [0, 0])
    assert len(ms) == 2
    ids = R.synthetic_inputs(1, 12, seed=5)[0]
    style = W.synthetic_voices(1)[0, 12, 0]
    ref = hip_model.infer([list(ids)], [list(style)], 1.0, seed=3)
    for m in ms:
        np.testing.assert_array_equal(m.infer([list(ids)], [list(style)], 1.0, seed=3), ref)
    d = hk.Dispatcher(ms, max_batch=4, max_wait_us=20000)
    out = [None] * 10
    errs = []

    def client(i):
        try:
            out[i] = d.submit(ids, style, 1.0, 40 + i)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=client, args=(i,)) for i in range(10)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    st = d.stats()
    d.close()
    assert not errs, errs
    assert st["requests"] == 10
    hip_model.set_utterance_base(0)
    for i in range(10):
        np.testing.assert_array_equal(out[i], hip_model.infer([list(ids)], [list(style)], 1.0, seed=40 + i))
    for m in ms:
        m.close()
    with pytest.raises(RuntimeError):
        hk.HipKoko.replicas(blob_path + ".missing", [0])


def test_32_clients_mixed_voices_and_formats_over_two_models(blob_path, hip_model, oracle):
    """BASELINE configs[4] as far as one GPU goes (kokorox-openai/src/lib.rs:370-439): 32 concurrent clients, the five
    voice rules of the serving benchmark incl. "af_sky.4+af_nicole.5" named into the device voice table, explicit style
    rows, all three output forms, two models behind one dispatcher.  Every request's bytes equal its solo call, both
    models take batches (a waiting queue is split between the idle workers), and one request is checked against the
    CPU oracle (not only against the HIP path itself)."""
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import voices as V
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R
    tab = W.synthetic_voices(4)
    names = ("af_sky", "af_nicole", "am_adam", "bf_emma")
    styles = {n: tab[i] for i, n in enumerate(names)}
    rules = [("af_sky", 0), ("af_nicole", 1), ("af_sky.4+af_nicole.5", [(0, 4.0), (1, 5.0)]), ("am_adam", 2),
             ("bf_emma.7+af_sky.3", [(3, 7.0), (0, 3.0)])]
    ms = hk.HipKoko.replicas(blob_path, [0, 0])
    for m in ms:
        m.set_voice_table(tab)
    hip_model.set_voice_table(tab)
    hip_model.set_utterance_base(0)
    n_clients, per = 32, 2
    reqs = {}
    for c in range(n_clients):
        for j in range(per):
            k = 6 + (7 * c + 3 * j) % 23
            ids = R.synthetic_inputs(1, k, seed=700 + 10 * c + j)[0]
            name, spec = rules[(c + j) % 5]
            by_row = (c % 4 == 3)  # a quarter of the clients send the 256-float row (the host-side mixer's result)
            reqs[(c, j)] = (ids, k, name, spec, by_row, (c + j) % 3, 5000 + 10 * c + j)
    d = hk.Dispatcher(ms, max_batch=16, max_wait_us=30000)
    out, errs = {}, []

    def client(c):
        try:
            for j in range(per):
                ids, k, name, spec, by_row, fmt, seed = reqs[(c, j)]
                if by_row:
                    out[(c, j)] = d.submit_ex(ids, style=V.mix_styles(styles, name, k)[0], seed=seed, fmt=fmt)
                else:
                    out[(c, j)] = d.submit_ex(ids, voices=spec, seed=seed, fmt=fmt)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=client, args=(c,)) for c in range(n_clients)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    st = d.stats()
    # error isolation: a request that is wrong on its own fails alone; the dispatcher keeps serving
    with pytest.raises(hk.KokoroxHipError, match="voice id"):
        d.submit_ex(reqs[(0, 0)][0], voices=99, seed=1)
    assert d.stats()["requests"] == st["requests"], "a bad voice id is refused at submit: it must never reach a batch"
    assert d.stats()["retried_batches"] == 0 and d.stats()["replayed_requests"] == 0, d.stats()  # (kx_dispatcher_failures)
    again = d.submit_ex(reqs[(0, 0)][0], voices=reqs[(0, 0)][3], seed=reqs[(0, 0)][6], fmt=reqs[(0, 0)][5])
    d.close()
    assert not errs, errs
    assert st["requests"] == n_clients * per and st["batches"] < n_clients * per and st["max_batch"] > 1
    assert all(b > 0 for b in st["batches_per_model"]), st  # both models worked
    np.testing.assert_array_equal(again, out[(0, 0)])
    for (c, j), (ids, k, name, spec, by_row, fmt, seed) in reqs.items():
        row = V.mix_styles(styles, name, k)[0]
        alone = hip_model.infer_packed([list(ids)], [row], [1.0], seed=seed, fmt=fmt)[0]
        np.testing.assert_array_equal(out[(c, j)], alone, err_msg=f"client {c} request {j} ({name}, fmt {fmt})")
    # one dispatched request against the oracle (f32 mono, a mix): durations exact, waveform under the usual protocol
    c0 = next(key for key, v in reqs.items() if v[5] == 0 and v[2] == "af_sky.4+af_nicole.5" and not v[4])
    ids, k, name, spec, by_row, fmt, seed = reqs[c0]
    row = np.asarray(V.mix_styles(styles, name, k)[0], dtype=np.float32)
    hip_model.infer([list(ids)], [list(row)], 1.0, seed=seed, flags=hk.KX_FLAG_TAPS)
    f0, n_c, har = hip_model.tap("pred.F0", 0), hip_model.tap("pred.N", 0)[0], hip_model.tap("gen.har", 0)
    audio, dur = oracle.forward(ids, row, 1.0, seed=seed, utt=0, f0_override=f0[0], n_override=n_c, har_override=har)
    assert out[c0].shape[0] == 600 * int(dur.sum())
    assert np.abs(out[c0] - audio.numpy()).max() < 1e-4
    for m in ms:
        m.close()


def test_results_are_copied_out_once_too_many_shared_buffers_are_held(blob_path):
    """Dispatcher results point into their batch's page-locked buffer; a client that keeps results keeps those buffers pinned.
    Beyond KX_PINNED_LIVE_CAP_MB of them the dispatcher copies a batch's results out instead.  With the cap at 0 every
    batch takes that path: same bytes as the same requests run alone, and the pool holds nothing afterwards (own process: the cap is
    read once)."""
    import os
    import subprocess
    import sys
    code = r"""
import threading, numpy as np
from kokorox_amd import hip_koko as hk
from kokorox_amd import weights as W
from oracle import kokoro_ref as R
m = hk.HipKoko.new(%r)
voices = W.synthetic_voices(4)
reqs = [(R.synthetic_inputs(1, 8 + 3 * i, seed=300 + i)[0], voices[i %% 4, 8 + 3 * i, 0], 1.0, 50 + i) for i in range(8)]
d = hk.Dispatcher([m], max_batch=8, max_wait_us=200000)
out = [None] * len(reqs)
def client(i):
    out[i] = d.submit(*reqs[i])
th = [threading.Thread(target=client, args=(i,)) for i in range(len(reqs))]
[t.start() for t in th]
[t.join(timeout=300) for t in th]
st = d.stats()
d.close()
assert st["batches"] < len(reqs), st
m.set_utterance_base(0)
for i, (ids, style, speed, seed) in enumerate(reqs):
    np.testing.assert_array_equal(out[i], m.infer([list(ids)], [list(style)], speed, seed=seed))
m.close()
print("copied out ok")
""" % blob_path
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, KX_PINNED_LIVE_CAP_MB="0"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "copied out ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
