#!/bin/bash
# phase stamps of a 128 -> 128, k = 11 launch: the 16x16x32 form on its 192-column tile against the 2 x 2-wave 32x32x16 form on 256
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  KX_DA_S16=$v KX_STAMP_K=11 KX_LIB=kokorox_amd/lib/variants/lib_stamps2.so KX_STAMP=gpurun_out/r03_st9_$v.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > /dev/null 2> gpurun_out/r03_st9.err || { tail -5 gpurun_out/r03_st9.err; exit 1; }
  echo "== KX_DA_S16=$v"; python tools/stamp_timeline.py gpurun_out/r03_st9_$v.bin 20 | head -6
done
