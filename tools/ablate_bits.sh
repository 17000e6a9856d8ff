#!/bin/bash
# run-time ablations of the one-role f16x3 conv on ONE box: tools/ablate_bits.sh <tag> <KX_DBG value>...
# (bits: 1 no input staging, 2 no weight copies, 4 no MFMAs, 8 no epilogue, 32 stores over row 0, 64 residual from row 0)
tag=$1; shift
for dbg in "$@"; do
  KX_DBG=$dbg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 \
      --detail gpurun_out/${tag}_dbg$dbg.txt > gpurun_out/${tag}_dbg$dbg.json 2> gpurun_out/${tag}_dbg$dbg.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_dbg$dbg.json").read().strip().splitlines()[-1])
rows = [l.split() for l in open("gpurun_out/${tag}_dbg$dbg.txt").read().splitlines()[1:]]
pick = {("128","128","11","1"): "k11", ("128","128","3","1"): "k3", ("256","256","7","1"): "k7x256", ("768","2048","1","1"): "gemm"}
extra = " ".join("%s %.2f" % (pick[(r[0],r[1],r[2],r[3])], float(r[8])) for r in rows if (r[0],r[1],r[2],r[3]) in pick)
print("KX_DBG=%-3s %.2f ms/step, conv avg %.4f ms | ms per step: %s" % ("$dbg", d["ms_per_step"], d["roofline"]["avg_launch_ms"], extra))
PY
done
