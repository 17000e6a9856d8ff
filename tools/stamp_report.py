"""Per-phase medians of the in-kernel stamps written by KX_STAMP=<file> (see Model::conv)."""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
us = lambda x: x / 100.0
st = [us(d[:, i] - t0) for i in range(4)]
print(f"{len(d)} workgroups, kernel span {st[3].max():.1f} us")
print(f"medians: first staging {np.median(st[1]-st[0]):.1f} us, main loop {np.median(st[2]-st[1]):.1f} us "
      f"(transform inside {np.median(us(d[:,6])):.1f}, barrier after transform {np.median(us(d[:,7] & np.uint64(0xffffffff))):.1f}, piece barriers {np.median(us(d[:,7] >> np.uint64(32))):.1f}), "
      f"epilogue {np.median(st[3]-st[2]):.1f} us, total {np.median(st[3]-st[0]):.1f} us")
T = st[3].max()
bins = np.arange(0, T + 2, 2.0)
occ = np.zeros(len(bins))
for s, e in zip(st[0], st[3]):
    occ[int(s // 2):int(e // 2) + 1] += 1
lo = len(bins) // 4
print(f"resident workgroups (middle half of the run): mean {occ[lo:3*lo].mean():.0f}")
clk = d[:, 5].astype(np.float64) / ((d[:, 3] - d[:, 0]).astype(np.float64) / 100.0)  # cycles per us = MHz
print(f"shader clock while the workgroups ran: median {np.median(clk):.0f} MHz (p10 {np.percentile(clk,10):.0f}, p90 {np.percentile(clk,90):.0f})")
