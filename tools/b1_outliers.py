"""Batch-1 latency outliers: N calls of one 128-phoneme utterance (kx_infer_device + kx_sync); prints the slowest calls with their
host milestones (kx_call_times: front queued / front done / back planned / back queued, then the total) so that a stall can be
placed: front half on the GPU (front_done late), host (front_queued late), back half (total - back_queued large).
    python tools/b1_outliers.py [calls=400] [gc=1]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
if len(sys.argv) > 2 and sys.argv[2] == "0":
    gc.disable()
dev = torch.device("cuda:0")
m = hk.HipKoko.new(W.ensure_synthetic_blob())
m.set_pinned_durations([3, 3, 3, 4])
T = 130
ids = torch.from_numpy(bench.synthetic_ids(1, 128, seed=1000)).to(dev)
st = torch.from_numpy(W.synthetic_voices(4)[0, 128, 0][None].copy()).to(dev)
F = int(np.array([3, 3, 3, 4] * 33)[:T].sum())
audio = torch.empty((1, 600 * F), dtype=torch.float32, device=dev)
fr = torch.zeros(1, dtype=torch.int32, device=dev)
lens = np.array([T], dtype=np.int32)
sp = np.ones(1, dtype=np.float32)
torch.cuda.synchronize()
rows = []
for i in range(n + 5):
    t = time.perf_counter()
    m.infer_device(ids.data_ptr(), T, lens, st.data_ptr(), sp, audio.data_ptr(), 600 * F, fr.data_ptr(), seed=2)
    t1 = time.perf_counter()
    m.sync()
    t2 = time.perf_counter()
    if i >= 5:
        rows.append([(t2 - t) * 1e3, (t1 - t) * 1e3] + m.call_times())
a = np.array(rows)
print(f"{n} calls (gc {'on' if gc.isenabled() else 'off'}, KX_LSTM_PARTS={os.environ.get('KX_LSTM_PARTS', '0')}): median {np.median(a[:, 0]):.3f} ms, p99 {np.sort(a[:, 0])[int(n * 0.99)]:.3f}, max {a[:, 0].max():.3f}; status {m.status()}")
for r in a[np.argsort(-a[:, 0])[:4]]:
    print("   total %.2f  returned %.2f | front queued %.2f  front done %.2f  back planned %.2f  back queued %.2f" % tuple(r))
m.close()
