"""Main-loop length of the stamped workgroups in SHADER CYCLES (clock-independent), from a -DKX_DA_STAMPS build:
usage: python tools/stamp_cycles.py <stamp file> <chunks per tile>"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 0] > 0]
n = int(sys.argv[2])
tot_us = (d[:, 3] - d[:, 0]).astype(np.float64) / 100.0
main_us = (d[:, 2] - d[:, 1]).astype(np.float64) / 100.0
mhz = d[:, 5].astype(np.float64) / tot_us
cyc = main_us * mhz
print(f"{len(d)} workgroups: clock p50 {np.median(mhz):.0f} MHz; main loop p50 {np.median(main_us):.1f} us = {np.median(cyc) / 1e3:.1f} k cycles "
      f"= {np.median(cyc) / n / 1e3:.2f} k cycles per chunk")
