#!/bin/bash
# build_variant_w2.sh NAME [-D flags...]: as build_variant.sh, for the W2 translation unit (conv_f16x3_da_w2.hip)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
L=$ROOT/kokorox_amd/lib; mkdir -p $L/variants /tmp/kxv
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -I$ROOT/kokorox_amd/csrc -c $ROOT/kokorox_amd/csrc/conv_f16x3_da_w2.hip -o /tmp/kxv/w2_$NAME.o 2>/tmp/kxv/w2_$NAME.err || { grep -w error -A3 /tmp/kxv/w2_$NAME.err | head; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/conv_mfma.o $L/conv_f16x3.o $L/conv_f16x3_da.o $L/conv_f16x3_da_p1.o /tmp/kxv/w2_$NAME.o $L/conv_f16x3_dag.o $L/kernels_misc.o $L/model.o $L/api.o $L/dispatcher.o -lpthread -o $L/variants/lib_$NAME.so
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -S --cuda-device-only "$@" -I$ROOT/kokorox_amd/csrc $ROOT/kokorox_amd/csrc/conv_f16x3_da_w2.hip -o /tmp/kxv/w2_$NAME.s 2>/dev/null
python3 /tmp/kxv/chk.py /tmp/kxv/w2_$NAME.s
