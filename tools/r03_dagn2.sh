#!/bin/bash
# narrow direct-A GEMM: where does it stop paying?  ms per step by batch with it off / on (grid rule) / forced everywhere
cd $GRAFT_REPO_ROOT
for bsz in 4 16 32 64; do
for v in 0 1 2; do
  KX_DAGN=$v timeout -k 10 300 python bench.py --batch $bsz --steps 6 --warmup 2 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 2> gpurun_out/r03_dagn2.err | { echo -n "batch $bsz KX_DAGN=$v: "; python tools/print_bench.py; } || { tail -5 gpurun_out/r03_dagn2.err; exit 1; }
done
done
