"""Time spent in the chunk barriers of the main loop (wave 0 of each workgroup), from the stamps of a -DKX_DA_STAMPS build:
usage: python tools/stamp_barrier.py <stamp file> <barriers per tile>"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 0] > 0]
nb = int(sys.argv[2])
bar = d[:, 7].astype(np.float64) / 100.0  # us
main = (d[:, 2] - d[:, 1]).astype(np.float64) / 100.0
print(f"{len(d)} workgroups: main loop p50 {np.median(main):.1f} us; in its {nb} chunk barriers p50 {np.median(bar):.2f} us "
      f"(p10 {np.percentile(bar, 10):.2f}, p90 {np.percentile(bar, 90):.2f}) = {100 * np.median(bar / main):.1f} % of the main loop, "
      f"{np.median(bar) / nb:.3f} us per barrier")
