#!/bin/bash
# Timing ablations of the conv kernels on one box (KX_DBG bits: 1 skip input transform, 2 skip weight copies,
# 4 skip MFMAs, 8 skip epilogue; results are then garbage, shapes are not: durations are pinned).
# usage: tools/ablate_ws.sh <tag> "<dbg values>"   -> gpurun_out/<tag>_dbg<N>.txt (per-shape tables)
tag=$1; shift
for d in $1; do
  KX_DBG=$d timeout -k 10 120 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 \
      --detail gpurun_out/${tag}_dbg${d}.txt > gpurun_out/${tag}_dbg${d}.json 2> gpurun_out/${tag}_dbg${d}.err || exit 1
  python3 - <<PY
import json
j=json.load(open("gpurun_out/${tag}_dbg${d}.json"))
print("dbg=${d}", "ms/step", round(j["ms_per_step"],2), "conv family TF/s", round(j["roofline"]["achieved"],1))
PY
done
