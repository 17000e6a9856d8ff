#!/bin/bash
# The 16x16x32 form of the direct-A conv (192- / 128-column tiles, 64-column statistics slots).  Parity against the LDS form
# (tolerance: one instruction sums 32 products), then A/B of the step with the form off / on.
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python - <<'PY' || exit 1
import numpy as np
from kokorox_amd import hip_koko as hk
rng = np.random.default_rng(11)
for (B, C, L, k, d) in ((2, 128, 517, 11, 5), (1, 256, 261, 7, 1), (3, 128, 3000, 7, 3), (8, 128, 20000, 11, 1), (4, 256, 6000, 7, 3), (1, 128, 25000, 7, 3)):
    x = rng.standard_normal((B, C, L), dtype=np.float32)
    w = (rng.standard_normal((C, C, k), dtype=np.float32) / np.sqrt(C * k)).astype(np.float32)
    b = rng.standard_normal(C, dtype=np.float32)
    alpha = (0.5 + rng.random(C)).astype(np.float32)
    norm = rng.standard_normal((B, 3, C), dtype=np.float32)
    norm[:, 1] = 1.0 + 0.1 * norm[:, 1]
    kw = dict(pad=d * (k - 1) // 2, dil=d, act=2, alpha=alpha, norm=norm)
    y2 = hk.conv1d(x, w, b, mode=2, **kw)
    y1 = hk.conv1d(x, w, b, mode=1, **kw)
    y3 = hk.conv1d(x, w, b, mode=3, **kw)
    e1, e3 = float(np.abs(y2 - y1).max()), float(np.abs(y2 - y3).max())
    print("S16 vs LDS form", (B, C, L, k, d), "mode 1 %.3e  mode 3 %.3e" % (e1, e3), flush=True)
    assert np.isfinite(y3).all() and e1 < 2e-5 and e3 < 2e-5 and (e1 > 0) == (k == 11)  # (7 taps: the 32x32x16 forms, bit-identical)
print("parity ok")
PY
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q -m gpu > gpurun_out/r03_s16_fwd.log 2>&1; tail -4 gpurun_out/r03_s16_fwd.log
for rep in 1 2; do
for v in 0 1; do
  KX_DA_S16=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --detail gpurun_out/r03_s16_${v}_$rep.txt 2> gpurun_out/r03_s16.err | { echo -n "KX_DA_S16=$v (round $rep): "; python tools/print_bench.py; } || { tail -5 gpurun_out/r03_s16.err; exit 1; }
done
done
