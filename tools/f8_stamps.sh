#!/bin/bash
# per-workgroup phase stamps of one 128 -> 128, k = 11 launch (diagnostic builds -DKX_DA_STAMPS of a translation unit:
# tools/build_variant_tu.sh), tools/stamp_timeline.py on each: tools/f8_stamps.sh "<KOKOROX_CONV> <variant>" ...
for arm in "$@"; do
  set -- $arm
  KOKOROX_CONV=$1 KX_LIB=kokorox_amd/lib/variants/lib_$2.so KX_STAMP=gpurun_out/stamps_$2.bin KX_STAMP_SKIP=9 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 > gpurun_out/stamps_$2.json 2> gpurun_out/stamps_$2.err || exit 1
  echo "== $1 $2"
  python tools/stamp_timeline.py gpurun_out/stamps_$2.bin 20 ${3:-} > gpurun_out/stamps_$2.txt 2>&1
  sed -n 1,6p gpurun_out/stamps_$2.txt; grep -E "shader clock|wave 0" gpurun_out/stamps_$2.txt
done
