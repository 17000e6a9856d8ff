"""f16f8 conv kernels against float64 and against the f16x3 kernels: error and (under rocprofv3) kernel time.
usage: python tools/f8_check.py  (on the GPU box)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokorox_amd import hip_koko as hk  # noqa: E402


def act_ref(x, norm, alpha):
    m, s, h = (torch.from_numpy(norm[i]).double()[:, :, None] for i in range(3))
    y = (x - m) * s + h
    al = torch.from_numpy(alpha).double()[None, :, None]
    return y + torch.sin(al * y) ** 2 / al


def main():
    rng = np.random.default_rng(5)
    for (B, C, Co, L, k, d) in [(1, 128, 128, 3000, 11, 1), (1, 128, 128, 3000, 11, 3), (1, 128, 128, 3000, 11, 5), (1, 256, 256, 2000, 7, 1),
                                (8, 128, 128, 25000, 11, 3), (8, 256, 256, 6000, 7, 1), (2, 56, 130, 517, 11, 5), (3, 32, 128, 65, 11, 1)]:
        x = rng.standard_normal((B, C, L), dtype=np.float32)
        w = (rng.standard_normal((Co, C, k), dtype=np.float32) / np.sqrt(C * k)).astype(np.float32)
        b = rng.standard_normal(Co, dtype=np.float32)
        alpha = (0.5 + rng.random(C)).astype(np.float32)
        norm = rng.standard_normal((3, B, C), dtype=np.float32)
        norm[1] = 1.0 + 0.1 * norm[1]
        kw = dict(pad=d * (k - 1) // 2, dil=d, act=2, alpha=alpha)
        y3 = hk.conv1d(x, w, b, norm=norm, mode=1, **kw)
        y8 = hk.conv1d(x, w, b, norm=norm, mode=1 | 0x200, **kw)
        nb = min(B, 2)
        xt = act_ref(torch.from_numpy(x[:nb]).double(), norm[:, :nb], alpha)
        ref = F.conv1d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=d * (k - 1) // 2, dilation=d).numpy()
        e3 = np.abs(y3[:nb] - ref).max()
        e8 = np.abs(y8[:nb] - ref).max()
        r8 = np.sqrt(((y8[:nb] - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())
        r3 = np.sqrt(((y3[:nb] - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())
        print(f"B {B} C {C}->{Co} L {L} k {k} d {d}: max|d| f16x3 {e3:.2e} f16f8 {e8:.2e}; rel rms f16x3 {r3:.2e} f16f8 {r8:.2e}; finite {np.isfinite(y8).all()}", flush=True)


if __name__ == "__main__":
    main()
