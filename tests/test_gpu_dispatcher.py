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
