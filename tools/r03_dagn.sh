#!/bin/bash
# narrow direct-A GEMM (conv1d_f16x3_dagn_kernel): parity, then batch-1 latency with and without it on one box
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > gpurun_out/r03_dagn_pytest.log 2>&1 || { tail -20 gpurun_out/r03_dagn_pytest.log; exit 1; }
tail -2 gpurun_out/r03_dagn_pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q -m gpu > gpurun_out/r03_dagn_fwd.log 2>&1 || { tail -20 gpurun_out/r03_dagn_fwd.log; exit 1; }
tail -2 gpurun_out/r03_dagn_fwd.log
for rep in 1 2; do
for v in 0 1; do
  KX_DAGN=$v timeout -k 10 200 python bench.py --batch 1 --steps 30 --warmup 5 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 2> gpurun_out/r03_dagn.err | { echo -n "KX_DAGN=$v batch 1 (round $rep): "; python tools/print_bench.py; } || { tail -5 gpurun_out/r03_dagn.err; exit 1; }
done
done
