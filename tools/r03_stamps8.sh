#!/bin/bash
# phase stamps of the polyphase upsampler launches (run-time tap form, scatter store): where does a tile's time go?
cd $GRAFT_REPO_ROOT
for rows in 768 2560; do
  KX_STAMP_K=2 KX_STAMP_ROWS=$rows KX_LIB=kokorox_amd/lib/variants/lib_stamps.so KX_STAMP=gpurun_out/r03_st8_$rows.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > /dev/null 2> gpurun_out/r03_st8.err || { tail -5 gpurun_out/r03_st8.err; exit 1; }
  echo "== rows $rows, K = 2"; python tools/stamp_timeline.py gpurun_out/r03_st8_$rows.bin 20 | head -8
done
