#!/bin/bash
# reduced-precision mode with the 2 x 2 wave layout: parity of the mode, then A/B
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "reduced or precision" > gpurun_out/r03_w2p1_pytest.log 2>&1 || { tail -30 gpurun_out/r03_w2p1_pytest.log; exit 1; }
tail -2 gpurun_out/r03_w2p1_pytest.log
for rep in 1 2; do
for v in 0 1; do
  KOKOROX_CONV=f16 KX_DA_W2=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 2> gpurun_out/r03_w2p1.err | { echo -n "f16 mode KX_DA_W2=$v (round $rep): "; python tools/print_bench.py; } || { tail -5 gpurun_out/r03_w2p1.err; exit 1; }
done
done
