#!/bin/bash
# round 3, run 7: more lanes at small batch (F0 / N branches, shortcut convs) -- forward tests + batch-1 A/B
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_dispatcher.py -x -q -m gpu > gpurun_out/r03i_pytest.log 2>&1; rc=$?
tail -8 gpurun_out/r03i_pytest.log
[ $rc -ne 0 ] && exit $rc
for l in 1 0 1 0; do
  KX_LANES=$l timeout -k 10 200 python bench.py --batch 1 --steps 20 --warmup 3 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > gpurun_out/r03i_b1_l$l.json 2> gpurun_out/r03i_b1_l$l.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03i_b1_l$l.json').read().strip().splitlines()[-1]); print('batch 1 lanes $l: %.3f ms/step' % d['ms_per_step'])"
done
for b in 4 16; do for l in 1 4; do
  KX_LANES=$l timeout -k 10 200 python bench.py --batch $b --steps 10 --warmup 2 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > gpurun_out/r03i_b${b}_l$l.json 2> gpurun_out/r03i_b${b}_l$l.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03i_b${b}_l$l.json').read().strip().splitlines()[-1]); print('batch $b lanes $l: %.3f ms/step' % d['ms_per_step'])"
done; done
