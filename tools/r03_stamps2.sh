#!/bin/bash
# phase stamps of the first 128 -> 128, k = 11 direct-A launch under the timing ablations (KX_DBG bits: 1 no input loads, 8 no epilogue)
cd $GRAFT_REPO_ROOT
for dbg in 0 8 1 9; do
  KX_DBG=$dbg KX_LIB=kokorox_amd/lib/variants/lib_stamps.so KX_STAMP=gpurun_out/r03_st_dbg$dbg.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > gpurun_out/r03_st_dbg$dbg.json 2> gpurun_out/r03_st_dbg$dbg.err || { tail -5 gpurun_out/r03_st_dbg$dbg.err; exit 1; }
  echo "== KX_DBG=$dbg"; python tools/stamp_timeline.py gpurun_out/r03_st_dbg$dbg.bin 5 > gpurun_out/r03_st_dbg$dbg.txt 2>/dev/null; sed -n 1,8p gpurun_out/r03_st_dbg$dbg.txt; grep "shader clock" gpurun_out/r03_st_dbg$dbg.txt
done
