#!/bin/bash
# r03_stamps7.sh NAME [k...]: cycles per chunk of the direct-A conv (in-kernel stamps) for kokorox_amd/lib/variants/lib_NAME.so
cd $GRAFT_REPO_ROOT
NAME=$1; shift
for k in ${@:-7 11}; do
  KX_STAMP_K=$k KX_LIB=kokorox_amd/lib/variants/lib_$NAME.so KX_STAMP=gpurun_out/r03_st7_${NAME}_k$k.bin timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > /dev/null 2> gpurun_out/r03_st7.err || { tail -5 gpurun_out/r03_st7.err; exit 1; }
  echo -n "$NAME k=$k: "; python tools/stamp_cycles.py gpurun_out/r03_st7_${NAME}_k$k.bin 8
done
