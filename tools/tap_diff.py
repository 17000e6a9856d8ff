"""Where does a batch member first differ from the same utterance run alone?  (taps of the forward, batch of 4 vs batch of 1)"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from kokorox_amd import hip_koko as hk
from kokorox_amd import weights as W
from test_gpu_forward import _inputs
m = hk.HipKoko.new(W.ensure_synthetic_blob())
counts = [17, 30, 12, 30]
ids, styles = _inputs(counts, seed0=40)
names = ["text_enc.out", "d_en", "dur.lstm", "pred.F0", "pred.N", "dec.encode", "dec.decode.3", "gen.har_source", "gen.har",
         "gen.x_source.0", "gen.ups.0", "gen.stage.0", "gen.x_source.1", "gen.ups.1", "gen.stage.1", "gen.conv_post", "audio"]
b = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m.set_utterance_base(0)
m.infer_batch([list(x) for x in ids], styles, [1.0], seed=9, flags=hk.KX_FLAG_TAPS)
tb = {}
for n in names:
    try:
        tb[n] = m.tap(n, b).copy()
    except Exception as e:
        tb[n] = None
m.set_utterance_base(b)
m.infer([list(ids[b])], [list(styles[b])], 1.0, seed=9, flags=hk.KX_FLAG_TAPS)
for n in names:
    if tb[n] is None:
        print(n, "no tap"); continue
    ta = m.tap(n, 0)
    L = min(ta.shape[1], tb[n].shape[1])
    d = np.abs(ta[:, :L] - tb[n][:, :L])
    bad = np.argwhere(d > 0)
    print(f"{n:16s} shape {ta.shape} vs {tb[n].shape}: max|d| {d.max():.3e}, differing {len(bad)}", "first", bad[:2].tolist(), "last", bad[-2:].tolist())
