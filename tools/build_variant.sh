#!/bin/bash
# Build an A/B variant of the library that differs only in conv_f16x3.hip's compile flags:
#   tools/build_variant.sh <name> [-DKX_...=...]...   ->  kokorox_amd/lib/variants/lib_<name>.so
# Select it at run time with KX_LIB=kokorox_amd/lib/variants/lib_<name>.so (hip_koko.load_library).  The other
# translation units are taken from the regular build (python -m kokorox_amd.build first).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
L=kokorox_amd/lib
mkdir -p $L/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-result "$@" \
    -c kokorox_amd/csrc/conv_f16x3.hip -o $L/variants/conv_f16x3_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/conv_mfma.o $L/variants/conv_f16x3_$name.o $L/conv_f16x3_da.o $L/conv_f16x3_dag.o \
    $L/kernels_misc.o $L/model.o $L/api.o $L/dispatcher.o -lpthread -o $L/variants/lib_$name.so
echo built $L/variants/lib_$name.so
