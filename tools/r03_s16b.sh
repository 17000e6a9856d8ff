#!/bin/bash
# S16 form at small batches (its 128-column tile replaces the 32x32x16 128-column tile for the 11-tap convs there): off / on
cd $GRAFT_REPO_ROOT
for bsz in 1 4 16; do
for rep in 1 2; do
for v in 0 1; do
  KX_DA_S16=$v timeout -k 10 300 python bench.py --batch $bsz --steps 12 --warmup 3 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 2> gpurun_out/r03_s16b.err | { echo -n "batch $bsz KX_DA_S16=$v (round $rep): "; python tools/print_bench.py; } || { tail -5 gpurun_out/r03_s16b.err; exit 1; }
done
done
done
