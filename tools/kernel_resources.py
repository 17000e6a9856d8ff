"""Compact register / scratch table of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(\w[\w ]*\w)(?: \[[\w/]+\])?: (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void kx::", "")
    if pat and pat not in n:
        continue
    print(f"{n:60s} vgpr {r.get('VGPRs', -1):3d} agpr {r.get('AGPRs', -1):3d} spill {r.get('VGPRs Spill', r.get('VGPR Spill', -1)):3d} "
          f"scratch {r.get('ScratchSize', -1):4d} occ {r.get('Occupancy', -1)} lds {r.get('LDS Size', -1)}")
