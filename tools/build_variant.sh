#!/bin/bash
# build_variant.sh NAME [extra hipcc -D flags...]: compile conv_f16x3_da.hip with the flags into kokorox_amd/lib/variants/lib_NAME.so
# (the other objects are reused from kokorox_amd/lib), and run the asm audit helpers on the same flags first.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
L=$ROOT/kokorox_amd/lib; mkdir -p $L/variants /tmp/kxv
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -I$ROOT/kokorox_amd/csrc -c $ROOT/kokorox_amd/csrc/conv_f16x3_da.hip -o /tmp/kxv/da_$NAME.o 2>/tmp/kxv/da_$NAME.err || { grep -w error /tmp/kxv/da_$NAME.err | head; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/conv_mfma.o $L/conv_f16x3.o /tmp/kxv/da_$NAME.o $L/conv_f16x3_da_p1.o $L/conv_f16x3_da_w2.o $L/conv_f16x3_dag.o $L/kernels_misc.o $L/model.o $L/api.o $L/dispatcher.o -lpthread -o $L/variants/lib_$NAME.so
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -S --cuda-device-only "$@" -I$ROOT/kokorox_amd/csrc $ROOT/kokorox_amd/csrc/conv_f16x3_da.hip -o /tmp/kxv/da_$NAME.s 2>/dev/null
python3 - /tmp/kxv/da_$NAME.s <<'P'
import re, sys, importlib.util, os
root=os.environ.get('KX_ROOT','/root/repo')
spec=importlib.util.spec_from_file_location('aud',root+'/tests/test_asm_audit_cpu.py'); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
ks=m._kernels(open(sys.argv[1]).read())
ok=True
for name,lines in ks.items():
    mm=re.search(r"da_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb[01]E", name); act,kt,ntt=(int(x) for x in mm.groups())
    bad=m._audit_no_touch_before_wait(lines); sp=any('scratch_' in l for l in lines)
    if bad or sp: ok=False; print('AUDIT', kt,ntt,act,'spill' if sp else '', 'bad',len(bad))
print('audit', 'clean' if ok else 'FAILED', len(ks),'kernels')
sys.exit(0 if ok else 1)
P
