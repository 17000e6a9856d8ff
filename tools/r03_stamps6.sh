#!/bin/bash
cd $GRAFT_REPO_ROOT
for k in 7 11; do
  KX_STAMP_K=$k KX_LIB=kokorox_amd/lib/variants/lib_stamps.so KX_STAMP=gpurun_out/r03_st6_k$k.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > /dev/null 2> gpurun_out/r03_st6.err || { tail -5 gpurun_out/r03_st6.err; exit 1; }
  echo -n "grouped regions k=$k: "; python tools/stamp_cycles.py gpurun_out/r03_st6_k$k.bin 8
done
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
tools/ab_variants.sh r03p main prev
