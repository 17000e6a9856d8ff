#!/bin/bash
# round 3, experiment 2 (one box): epilogue access-pattern probe; lanes at batch 1
cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/probes/epi_pattern.bin 88 > gpurun_out/r03b_epi_pattern.txt 2>&1 || { cat gpurun_out/r03b_epi_pattern.txt; exit 1; }
cat gpurun_out/r03b_epi_pattern.txt
timeout -k 10 120 tools/probes/epi_pattern.bin 24 > gpurun_out/r03b_epi_pattern_k3.txt 2>&1
cat gpurun_out/r03b_epi_pattern_k3.txt
for l in 1 4 1 4; do
  KX_LANES=$l timeout -k 10 200 python bench.py --batch 1 --steps 20 --warmup 3 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > gpurun_out/r03b_b1_l$l.json 2> gpurun_out/r03b_b1_l$l.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03b_b1_l$l.json').read().strip().splitlines()[-1]); print('batch 1 lanes $l: %.3f ms/step' % d['ms_per_step'])"
done
