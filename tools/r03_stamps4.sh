#!/bin/bash
# main-loop cycles with and without the interleaved transform (diagnostic builds), k = 3 / 7 / 11
cd $GRAFT_REPO_ROOT
for lib in stamps_nox; do for k in 3 7 11; do
  KX_STAMP_K=$k KX_LIB=kokorox_amd/lib/variants/lib_$lib.so KX_STAMP=gpurun_out/r03_st4_${lib}_k$k.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > /dev/null 2> gpurun_out/r03_st4.err || { tail -5 gpurun_out/r03_st4.err; exit 1; }
  echo -n "$lib k=$k: "; python tools/stamp_cycles.py gpurun_out/r03_st4_${lib}_k$k.bin 8
done; done
