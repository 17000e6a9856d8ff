#!/bin/bash
# phase stamps of the first 128 -> 128 direct-A launch with k = 3 and k = 7 (diagnostic build lib_stamps.so)
cd $GRAFT_REPO_ROOT
for k in 3 7 11; do
  KX_STAMP_K=$k KX_LIB=kokorox_amd/lib/variants/lib_stamps.so KX_STAMP=gpurun_out/r03_st_k$k.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > gpurun_out/r03_st_k$k.json 2> gpurun_out/r03_st_k$k.err || { tail -5 gpurun_out/r03_st_k$k.err; exit 1; }
  echo "== k = $k"; python tools/stamp_timeline.py gpurun_out/r03_st_k$k.bin 5 > gpurun_out/r03_st_k$k.txt 2>/dev/null; sed -n 1,6p gpurun_out/r03_st_k$k.txt; grep "shader clock" gpurun_out/r03_st_k$k.txt
done
