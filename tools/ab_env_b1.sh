#!/bin/bash
# A/B of run-time switches on ONE box with the regular library: tools/ab_env.sh <tag> "VAR=a" "VAR=b" ...  (two rounds)
tag=$1; shift
for round in 1 2; do
  i=0
  for kv in "$@"; do
    i=$((i+1))
    env $kv timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 40 \
        --detail gpurun_out/${tag}_${i}_$round.txt > gpurun_out/${tag}_${i}_$round.json 2> gpurun_out/${tag}_${i}_$round.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_${i}_$round.json").read().strip().splitlines()[-1])
print("$kv round $round: %.2f ms/step, conv avg %.4f ms, b1 median %.3f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["latency_b1"]["median_ms"]))
PY
  done
done
