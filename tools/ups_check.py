"""Polyphase transposed conv with a leaky input (the model's upsamplers) through the test hook, against torch float64."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from kokorox_amd import hip_koko as hk
rng = np.random.default_rng(11)
for (B, Cin, Cout, L, k, s) in [(8, 512, 256, 8440, 20, 10), (8, 256, 128, 20000, 12, 6), (2, 256, 128, 3000, 12, 6)]:
    x = rng.standard_normal((B, Cin, L), dtype=np.float32)
    w = (rng.standard_normal((Cin, Cout, k), dtype=np.float32) / np.sqrt(Cin * k)).astype(np.float32)
    b = rng.standard_normal(Cout, dtype=np.float32)
    p = (k - s) // 2
    y = hk.conv1d(x, w, b, stride=s, pad=p, transposed=True, act=1, slope=0.1, mode=1)
    ref = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x).double(), 0.1), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                             stride=s, padding=p).numpy()
    print((B, Cin, Cout, L, k, s), "max|d| vs f64:", float(np.abs(y - ref).max()), "finite:", bool(np.isfinite(y).all()))
    assert np.abs(y - ref).max() < 3e-5
print("ups ok")
