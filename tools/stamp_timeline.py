"""Chip-level timeline from the per-workgroup stamps of one conv launch (KX_STAMP=<file>, see Model::conv):
how many workgroups are in their first staging / main loop / epilogue at each instant, and how the phase lengths are
distributed.  Answers whether the epilogues (the HBM write + residual read bursts) bunch up in time.
usage: python tools/stamp_timeline.py <stamp file> [bin_us]"""
import sys
import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 0] > 0]
binw = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
t0 = d[:, 0].min()
st = [(d[:, i] - t0).astype(np.float64) / 100.0 for i in range(4)]  # us
span = st[3].max()
print(f"{len(d)} workgroups, kernel span {span:.1f} us")
for name, a, b in (("first staging", 0, 1), ("main loop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
    x = st[b] - st[a]
    print(f"  {name:14s} p10 {np.percentile(x, 10):7.1f}  p50 {np.median(x):7.1f}  p90 {np.percentile(x, 90):7.1f}  mean {x.mean():7.1f} us")
nb = int(span / binw) + 1


def occupancy(a, b):
    occ = np.zeros(nb)
    for s, e in zip(st[a], st[b]):
        i0, i1 = int(s / binw), int(e / binw)
        if i0 == i1:
            occ[i0] += (e - s) / binw
        else:
            occ[i0] += ((i0 + 1) * binw - s) / binw
            occ[i0 + 1:i1] += 1
            occ[i1] += (e - i1 * binw) / binw
    return occ


res = occupancy(0, 3)
epi = occupancy(2, 3)
stg = occupancy(0, 1)
lo, hi = nb // 8, nb - nb // 8  # steady part of the launch
print(f"steady part ({lo * binw:.0f}..{hi * binw:.0f} us): resident {res[lo:hi].mean():.0f} workgroups; "
      f"in epilogue mean {epi[lo:hi].mean():.1f} (std {epi[lo:hi].std():.1f}, min {epi[lo:hi].min():.0f}, max {epi[lo:hi].max():.0f}); "
      f"in first staging mean {stg[lo:hi].mean():.1f} (std {stg[lo:hi].std():.1f})")
# a Poisson-like (unbunched) process has std ~ sqrt(mean); convoys show as std >> sqrt(mean)
print(f"  sqrt(mean) = {np.sqrt(epi[lo:hi].mean()):.1f}")
# epilogue length against how many others were in their epilogue at its start
k = (st[2] / binw).astype(int).clip(0, nb - 1)
load = epi[k]
el = st[3] - st[2]
for q0, q1 in ((0, 25), (25, 50), (50, 75), (75, 100)):
    a, b = np.percentile(load, q0), np.percentile(load, q1)
    m = (load >= a) & (load <= b)
    print(f"  epilogues that start with {a:5.1f}..{b:5.1f} others in epilogue: median length {np.median(el[m]):6.1f} us (n = {m.sum()})")
clk = d[:, 5].astype(np.float64) / (st[3] - st[0])
print(f"shader clock: median {np.median(clk):.0f} MHz")
print("timeline (bin start us: resident / staging / epilogue):")
step = max(1, nb // 40)
for i in range(0, nb, step):
    print(f"  {i * binw:7.0f}: {res[i]:5.0f} {stg[i]:6.1f} {epi[i]:6.1f}")
if len(sys.argv) > 3 and sys.argv[3] == "drain":  # KX_DBG=128: o[6] = stamp taken before the final s_waitcnt of wave 0
    pre = (d[:, 6] - t0).astype(np.float64) / 100.0
    x = st[3] - pre
    y = pre - st[2]
    print(f"epilogue issue part p50 {np.median(y):.1f} us (p10 {np.percentile(y,10):.1f}, p90 {np.percentile(y,90):.1f}); "
          f"drain of the stores after the last one was issued p50 {np.median(x):.1f} us (p10 {np.percentile(x,10):.1f}, p90 {np.percentile(x,90):.1f})")
if len(sys.argv) > 3 and sys.argv[3] == "f8":  # F8 forms: o[7] = 10 ns ticks of wave 0 in the loop's barriers, o[6] = in its ring waits
    bar = d[:, 7].astype(np.float64) / 100.0
    wt = d[:, 6].astype(np.float64) / 100.0
    loop = st[2] - st[1]
    print(f"wave 0, main loop: in barriers p50 {np.median(bar):.1f} us ({100 * np.median(bar / loop):.1f} % of the loop), in ring waits p50 {np.median(wt):.1f} us "
          f"({100 * np.median(wt / loop):.1f} %)")
