"""BASELINE config 5 on one GPU: N concurrent clients, mixed voices incl. "af_sky.4+af_nicole.5", requests
coalesced by the dispatcher.  Prints one JSON line (p50/p99 latency, aggregate RTF, batch statistics)."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import voices as V  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=32)
    ap.add_argument("--requests", type=int, default=8, help="per client")
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--max-wait-us", type=int, default=3000)
    ap.add_argument("--serial", action="store_true", help="reference behaviour: one request at a time (max_batch 1)")
    a = ap.parse_args()
    m = hk.HipKoko.new(W.ensure_synthetic_blob())
    tab = W.synthetic_voices(4)
    styles = {n: tab[i] for i, n in enumerate(("af_sky", "af_nicole", "am_adam", "bf_emma"))}
    names = ["af_sky", "af_nicole", "af_sky.4+af_nicole.5", "am_adam", "bf_emma.7+af_sky.3"]
    d = hk.Dispatcher([m], max_batch=1 if a.serial else a.max_batch, max_wait_us=0 if a.serial else a.max_wait_us)
    rng = np.random.default_rng(0)
    plan = [[(int(rng.integers(20, 129)), names[int(rng.integers(0, len(names)))], int(rng.integers(1, 2 ** 31)))
             for _ in range(a.requests)] for _ in range(a.clients)]
    # warm-up: one batch of the largest shape so the arenas exist
    d.submit([0] + [5] * 128 + [0], V.mix_styles(styles, "af_sky", 128)[0], 1.0, 1)
    lat, audio = [], []
    lock = threading.Lock()

    def client(c):
        r = np.random.default_rng(100 + c)
        for (k, voice, seed) in plan[c]:
            ids = np.concatenate([[0], r.integers(1, 178, size=k), [0]]).astype(np.int64)
            style = V.mix_styles(styles, voice, k)[0]   # host-side mixer, koko.rs:1255-1306
            t = time.perf_counter()
            w = d.submit(ids, style, 1.0, seed)
            dt = time.perf_counter() - t
            with lock:
                lat.append(dt)
                audio.append(w.shape[0] / 24000.0)

    t0 = time.perf_counter()
    th = [threading.Thread(target=client, args=(c,)) for c in range(a.clients)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    st = d.stats()
    d.close()
    m.close()
    lat = np.sort(np.array(lat))
    print(json.dumps({"clients": a.clients, "requests": len(lat), "serial": bool(a.serial), "wall_s": wall,
                      "audio_s": float(np.sum(audio)), "aggregate_rtf": float(np.sum(audio) / wall),
                      "latency_p50_ms": float(lat[len(lat) // 2] * 1e3), "latency_p99_ms": float(lat[int(len(lat) * 0.99)] * 1e3),
                      "batches": st["batches"] - 1, "max_batch": st["max_batch"], "requests_per_s": len(lat) / wall}))


if __name__ == "__main__":
    main()
