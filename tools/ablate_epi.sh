#!/bin/bash
# epilogue ablations of the one-role f16x3 conv on one box (KX_DBG bits: 8 no epilogue, 32 stores over row 0, 64 residual from row 0)
for dbg in 0 8 32 96 0; do
  KX_DBG=$dbg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 \
      --detail gpurun_out/epi_dbg$dbg.txt > gpurun_out/epi_dbg$dbg.json 2> gpurun_out/epi_dbg$dbg.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/epi_dbg$dbg.json").read().strip().splitlines()[-1])
print("KX_DBG=$dbg: %.2f ms/step, conv avg %.4f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
done
