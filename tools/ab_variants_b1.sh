#!/bin/bash
# A/B of library variants on ONE box: tools/ab_variants.sh <tag> <variant>...   ("main" = the regular library)
# Each arm: bench.py (3 steps) with the per-shape conv table -> gpurun_out/<tag>_<variant>.{json,txt}; two rounds, so that
# drift of the box shows.
tag=$1; shift
for round in 1 2; do
  for v in "$@"; do
    lib=kokorox_amd/lib/variants/lib_$v.so
    [ "$v" = main ] && lib=kokorox_amd/lib/libkokorox_hip.so
    KX_LIB=$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 40 \
        --detail gpurun_out/${tag}_${v}_$round.txt > gpurun_out/${tag}_${v}_$round.json 2> gpurun_out/${tag}_${v}_$round.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_${v}_$round.json").read().strip().splitlines()[-1])
print("$v round $round: %.2f ms/step, conv avg %.4f ms, b1 median %.3f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["latency_b1"]["median_ms"]))
PY
  done
done
