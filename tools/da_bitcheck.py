"""Bit comparison of the direct-A kernel (test-hook mode 3) with the LDS-weights kernel (mode 1) on launch-sized shapes."""
import sys
import numpy as np
sys.path.insert(0, ".")
from kokorox_amd import hip_koko as hk
rng = np.random.default_rng(5)
for (B, Cin, Cout, L, k, d) in [(8, 128, 128, 20000, 11, 1), (4, 256, 256, 6000, 7, 3), (8, 128, 128, 20000, 3, 5), (2, 128, 128, 3001, 11, 5)]:
    x = rng.standard_normal((B, Cin, L), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k), dtype=np.float32) / np.sqrt(Cin * k)).astype(np.float32)
    b = rng.standard_normal(Cout, dtype=np.float32)
    alpha = (0.5 + rng.random(Cin)).astype(np.float32)
    norm = rng.standard_normal((B, 3, Cin), dtype=np.float32)
    norm[:, 1] = 1.0 + 0.1 * norm[:, 1]
    p = d * (k - 1) // 2
    for kw in (dict(), dict(act=2, alpha=alpha, norm=norm)):
        y1 = hk.conv1d(x, w, b, pad=p, dil=d, mode=1, **kw)
        y3 = hk.conv1d(x, w, b, pad=p, dil=d, mode=3, **kw)
        bad = np.argwhere(y1 != y3)
        print((B, Cin, Cout, L, k, d), "act" if kw else "plain", "mismatches:", len(bad), "max|d|", float(np.abs(y1 - y3).max()),
              "first", bad[:3].tolist())

# epilogue forms and the polyphase transposed conv, at a batch where mode 1 picks the small-grid 128 x (4 x 1) tile
for (B, Cin, Cout, L, k, d) in [(1, 128, 128, 5000, 7, 1), (3, 256, 256, 2100, 3, 1), (8, 128, 128, 20000, 7, 1)]:
    x = rng.standard_normal((B, Cin, L), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k), dtype=np.float32) / np.sqrt(Cin * k)).astype(np.float32)
    b = rng.standard_normal(Cout, dtype=np.float32)
    p = d * (k - 1) // 2
    res = rng.standard_normal((B, Cout, L), dtype=np.float32)
    run = rng.standard_normal((B, Cout, L), dtype=np.float32)
    for name, kw in (("resid+stats", dict(resid=res, want_stats=True)), ("resid+accum+div", dict(resid=res, y_init=run, out_div=3.0)),
                     ("mul+stats", dict(out_mul=0.7071, want_stats=True))):
        o1 = hk.conv1d_epilogue(x, w, b, pad=p, dil=d, mode=1, **kw)
        o3 = hk.conv1d_epilogue(x, w, b, pad=p, dil=d, mode=3, **kw)
        if isinstance(o1, tuple):
            print((B, Cin, Cout, L, k, d), name, "y mismatches:", int((o1[0] != o3[0]).sum()), "stats mismatches:", int((o1[1] != o3[1]).sum()),
                  "max|d stats|", float(np.abs(o1[1] - o3[1]).max()))
        else:
            print((B, Cin, Cout, L, k, d), name, "y mismatches:", int((o1 != o3).sum()))
for (B, Cin, Cout, L, k, s) in [(1, 256, 128, 3000, 12, 6), (4, 512, 256, 900, 20, 10)]:
    x = rng.standard_normal((B, Cin, L), dtype=np.float32)
    w = (rng.standard_normal((Cin, Cout, k), dtype=np.float32) / np.sqrt(Cin * k)).astype(np.float32)
    b = rng.standard_normal(Cout, dtype=np.float32)
    y1 = hk.conv1d(x, w, b, stride=s, pad=(k - s) // 2, transposed=True, mode=1)
    y3 = hk.conv1d(x, w, b, stride=s, pad=(k - s) // 2, transposed=True, mode=3)
    print("transposed", (B, Cin, Cout, L, k, s), "mismatches:", int((y1 != y3).sum()), "max|d|", float(np.abs(y1 - y3).max()))

# k = 1 GEMMs: direct-A GEMM kernel (mode 1, the default) against the virtual-tap LDS-DMA form (mode 2 keeps it)
for (B, Cin, Cout, L) in [(4, 768, 2048, 2100), (2, 2048, 768, 4000), (8, 640, 2048, 130), (3, 1090, 1024, 845)]:
    x = rng.standard_normal((B, Cin, L), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, 1), dtype=np.float32) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout, dtype=np.float32)
    for kw in (dict(), dict(act=1, slope=0.2)):
        y1 = hk.conv1d(x, w, b, mode=1, **kw)
        y2 = hk.conv1d(x, w, b, mode=2, **kw)
        ref = np.einsum("oc,bcl->bol", w[:, :, 0].astype(np.float64), np.where(x > 0, x, x * kw.get("slope", 1.0)).astype(np.float64) if kw else x.astype(np.float64)) + b[None, :, None]
        print("gemm", (B, Cin, Cout, L), "leaky" if kw else "plain", "mismatches vs virtual-tap form:", int((y1 != y2).sum()),
              "max|d| vs f64:", float(np.abs(y1 - ref).max()))
