#!/bin/bash
# does the transform hide when it is spread over ALL tiles of a chunk?  Both arms WITHOUT input loads (KX_DBG=1: the transform
# runs on stale registers, nothing to wait for), main-loop CYCLES: default start tile vs start tile 0 (diagnostic builds)
cd $GRAFT_REPO_ROOT
for lib in stamps stamps_i0; do for k in 3 7 11; do
  KX_DBG=1 KX_STAMP_K=$k KX_LIB=kokorox_amd/lib/variants/lib_$lib.so KX_STAMP=gpurun_out/r03_st5_${lib}_k$k.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > /dev/null 2> gpurun_out/r03_st5.err || { tail -5 gpurun_out/r03_st5.err; exit 1; }
  echo -n "$lib k=$k (KX_DBG=1): "; python tools/stamp_cycles.py gpurun_out/r03_st5_${lib}_k$k.bin 8
done; done
