import numpy as np, sys
sys.path.insert(0, '/root/repo')
from kokorox_amd import hip_koko as hk
B, Cin, Cout, L, k, p = 1, 16, 128, 256, 3, 1
x = np.zeros((B, Cin, L), np.float32)
w = np.zeros((Cout, Cin, k), np.float32)
res = (np.arange(Cout)[:, None] * 1000 + np.arange(L)[None, :]).astype(np.float32)[None]
y = hk.conv1d_epilogue(x, w, None, pad=p, resid=res, mode=1)
print("max err", np.abs(y - res).max())
bad = np.argwhere(np.abs(y - res) > 0.5)
print("n bad", len(bad), bad[:10].tolist())
print("row0", y[0, 0, :12])
print("row1", y[0, 1, :12])
print("row9", y[0, 9, :12])
