#!/bin/bash
# closed-loop serving leg with 1 / 2 / 3 models on GPU 0 (repeated ids = CU-partitioned models), two rounds:
#   bash tools/serve_ab.sh TAG [models-arg ...]      default: "0" "0,0" "0,0,0"
tag=$1; shift
arms=("$@"); [ ${#arms[@]} -eq 0 ] && arms=("0" "0,0" "0,0,0")
for round in 1 2; do
  for m in "${arms[@]}"; do
    n=$(echo $m | tr -cd ',' | wc -c)
    timeout -k 10 300 python tools/serve_bench.py --models $m --open-loop-s 0 > gpurun_out/${tag}_m$((n+1))_$round.json 2> gpurun_out/${tag}_m$((n+1))_$round.err || { tail -5 gpurun_out/${tag}_m$((n+1))_$round.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_m$((n+1))_$round.json").read().strip().splitlines()[-1])
print("models $m round $round: %.0fx aggregate, p50 %.1f ms, p99 %.1f ms, %d batches (max %d), per model %s, retried %d" % (
    d["aggregate_rtf"], d["latency_p50_ms"], d["latency_p99_ms"], d["batches"], d["max_batch"], d["batches_per_model"], d["retried_batches"]))
PY
  done
done
