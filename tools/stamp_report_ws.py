"""Per-phase medians of the stamps the wave-specialised conv kernel writes with KX_STAMP=<file> (Model::conv picks the
first launch with KX_STAMP_ROWS rows and KX_STAMP_K taps; consumer wave 0, second tile of every workgroup)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
prod = raw[(raw[:, 0] == 1) & (raw[:, 7] > 0)]
d = raw[raw[:, 0] > 1]
us = lambda x: x.astype(np.float64) / 100.0
pro, main, epi = us(d[:, 1] - d[:, 0]), us(d[:, 2] - d[:, 1]), us(d[:, 3] - d[:, 2])
steps = (d[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
clk = d[:, 4].astype(np.float64) / main  # cycles per us = MHz
print(f"{len(d)} workgroups stamped; steps per tile {int(np.median(steps))}")
print(f"medians per tile: wait for the prologue {np.median(pro):.1f} us, main loop {np.median(main):.1f} us, "
      f"epilogue {np.median(epi):.1f} us (p10 {np.percentile(epi,10):.1f}, p90 {np.percentile(epi,90):.1f})")
print(f"shader clock in the main loop: median {np.median(clk):.0f} MHz (p10 {np.percentile(clk,10):.0f}, p90 {np.percentile(clk,90):.0f})")
cyc = d[:, 4].astype(np.float64)
print(f"main loop: {np.median(cyc):.0f} cycles = {np.median(cyc/steps):.0f} per step; inside step barriers "
      f"{np.median(d[:,6].astype(np.float64)/steps):.0f} per step ({100*np.median(d[:,6].astype(np.float64)/cyc):.1f} %)")
t0 = d[:, 0].min()
ep0 = us(d[:, 2] - t0)
print(f"epilogue start times: spread p10..p90 = {np.percentile(ep0,10):.1f}..{np.percentile(ep0,90):.1f} us after the first stamp")
if len(prod):
    nch = (prod[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
    npc = (prod[:, 7] >> np.uint64(32)).astype(np.float64)
    m = lambda x: np.median(x)
    f = lambda i: prod[:, i].astype(np.float64)
    print(f"producer wave 0: first step of a chunk {m(f(1)/nch):.0f} cycles (weight piece {m(f(3)/(nch*npc)):.0f}, transform + loads "
          f"{m(f(4)/nch):.0f}, barrier {m(f(5)/nch):.0f}); other steps {m(f(2)/np.maximum(nch*(npc-1),1)):.0f} cycles "
          f"(barrier {m(f(6)/np.maximum(nch*(npc-1),1)):.0f})")
