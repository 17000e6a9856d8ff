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
