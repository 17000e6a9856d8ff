"""BASELINE config 5 outside the bench line: N concurrent clients, mixed voices incl. "af_sky.4+af_nicole.5" named into the
device voice table, three output forms, requests coalesced by the dispatcher over one or more models.  Prints one JSON line
(p50/p99 latency, aggregate RTF, batch statistics) -- the same leg bench.py runs as its `serve` block (bench.serve_leg).

    python tools/serve_bench.py [--clients 32] [--requests 4] [--models 0,0] [--serial]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=32)
    ap.add_argument("--requests", type=int, default=48, help="per client (closed loop)")
    ap.add_argument("--open-loop-s", type=float, default=5.0, help="seconds per open-loop load (0 = closed loop only)")
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--max-wait-us", type=int, default=3000)
    ap.add_argument("--models", default="0", help="device ids, one model each (e.g. 0,0 = two models on one GPU)")
    ap.add_argument("--serial", action="store_true", help="reference behaviour: one request at a time (max_batch 1)")
    a = ap.parse_args()
    ids = [int(v) for v in a.models.split(",")]
    blob = W.ensure_synthetic_blob()
    models = hk.HipKoko.replicas(blob, ids) if len(ids) > 1 else [hk.HipKoko.new(blob, device=ids[0])]
    out = bench.serve_leg(models, n_clients=a.clients, per_client=a.requests, max_batch=1 if a.serial else a.max_batch,
                          max_wait_us=0 if a.serial else a.max_wait_us, open_loop_s=a.open_loop_s,
                          open_loads=(0.5, 0.75, 0.9) if a.open_loop_s > 0 else ())
    out["serial"] = bool(a.serial)
    print(json.dumps(out))
    for m in models:
        m.close()


if __name__ == "__main__":
    main()
