"""Two models on GPU 0, each running batch-32 forwards of 128-phoneme utterances from its own thread; every result must equal a
quiet run bit for bit and no call may fail.  Prints one JSON line {"ok", "errors", "status": [kx_model_status of each], "seconds"}.
    python tools/two_models_side_by_side.py partition|whole [rounds]
(whole + KX_DEVICE_TURN=0 in the environment = whole-device models NOT taking turns: what the library's re-run after a hand-off
time-out has to absorb; tests/test_gpu_kernels.py runs all three.)"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402  (test infrastructure: the synthetic inputs)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "partition"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    blob = W.ensure_synthetic_blob()
    B, n_ph = 32, 128
    toks = [list(R.synthetic_inputs(1, n_ph, seed=900 + i)[0]) for i in range(B)]
    voices = W.synthetic_voices(4)
    styles = [voices[i % 4, n_ph, 0] for i in range(B)]
    ms = hk.HipKoko.replicas(blob, [0, 0]) if mode == "partition" else [hk.HipKoko.new(blob), hk.HipKoko.new(blob)]
    errors, wrong = [], 0
    t0 = time.perf_counter()
    try:
        for m in ms:
            m.set_pinned_durations([3, 3, 3, 4])
        quiet = ms[0].infer_batch(toks, styles, [1.0], seed=11)
        res = [[], []]

        def run(i):
            try:
                for _ in range(rounds):
                    res[i].append(ms[i].infer_batch(toks, styles, [1.0], seed=11))
            except Exception as e:  # pragma: no cover
                errors.append(repr(e))

        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        for i in range(2):
            for out in res[i]:
                for a, b in zip(out, quiet):
                    wrong += 0 if np.array_equal(a, b) else 1
        status = [m.status() for m in ms]
    finally:
        for m in ms:
            m.close()
    print(json.dumps({"ok": not errors and wrong == 0 and all(len(r) == rounds for r in res), "errors": errors, "wrong": wrong, "status": status,
                      "seconds": time.perf_counter() - t0, "mode": mode, "device_turn": os.environ.get("KX_DEVICE_TURN", "1")}))


if __name__ == "__main__":
    main()
