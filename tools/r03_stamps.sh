#!/bin/bash
# per-workgroup phase stamps of the first 128 -> 128, k = 11 direct-A launch (diagnostic build lib_stamps.so), plain and de-phased
cd $GRAFT_REPO_ROOT
for cfg in "base" "deph KX_DEPHASE=500 KX_DEPHASE_MODE=2" "deph1 KX_DEPHASE=500 KX_DEPHASE_MODE=1"; do
  set -- $cfg; tag=$1; shift
  env "$@" KX_LIB=kokorox_amd/lib/variants/lib_stamps.so KX_STAMP=gpurun_out/r03_st_$tag.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > gpurun_out/r03_st_$tag.json 2> gpurun_out/r03_st_$tag.err || { tail -5 gpurun_out/r03_st_$tag.err; exit 1; }
  echo "== $cfg"; python tools/stamp_timeline.py gpurun_out/r03_st_$tag.bin 5 | head -24
done
