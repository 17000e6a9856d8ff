#!/bin/bash
# W2 form of the direct-A conv (2 x 2 waves): kernel parity first, then A/B against the 4 x 1 form on one box
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r03_w2_pytest.log 2>&1 || { tail -30 gpurun_out/r03_w2_pytest.log; exit 1; }
tail -2 gpurun_out/r03_w2_pytest.log
for rep in 1 2; do
for v in 0 1; do
  KX_DA_W2=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --detail gpurun_out/r03_w2_${v}_$rep.txt 2> gpurun_out/r03_w2.err | { echo -n "KX_DA_W2=$v (round $rep): "; python tools/print_bench.py; } || { tail -5 gpurun_out/r03_w2.err; exit 1; }
done
done
