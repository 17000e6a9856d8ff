#!/bin/bash
# build_variant_tu.sh NAME TU ALT_SOURCE [extra flags]: a library variant in which translation unit TU (e.g. kernels_misc) is compiled
# from ALT_SOURCE (e.g. an older revision written to /tmp by `git show REV:kokorox_amd/csrc/TU.hip`), every other object reused
# from kokorox_amd/lib -> kokorox_amd/lib/variants/lib_NAME.so (select with KX_LIB; tools/ab_variants.sh)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; TU=$2; ALT=$3; shift 3
L=$ROOT/kokorox_amd/lib; mkdir -p $L/variants /tmp/kxv
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -I$ROOT/kokorox_amd/csrc -I$ROOT/include -c $ALT -o /tmp/kxv/${TU}_$NAME.o
OBJS=""
for o in $L/*.o; do
  [ "$(basename $o)" = "$TU.o" ] && OBJS="$OBJS /tmp/kxv/${TU}_$NAME.o" || OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -lpthread -o $L/variants/lib_$NAME.so
echo built $L/variants/lib_$NAME.so
