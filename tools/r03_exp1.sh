#!/bin/bash
# round 3, experiment 1 (one box): lanes + de-phasing A/B, then the forward tests.
cd $GRAFT_REPO_ROOT
tools/ab_env.sh r03a "KX_LANES=1" "KX_LANES=4" "KX_LANES=1 KX_DEPHASE=500" "KX_LANES=1 KX_DEPHASE=500 KX_DEPHASE_MODE=2" "KX_LANES=4 KX_DEPHASE=500" "KX_LANES=1 KX_DEPHASE=250" || exit 1
timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -x -q -m gpu > gpurun_out/r03a_pytest.log 2>&1 || { tail -30 gpurun_out/r03a_pytest.log; exit 1; }
tail -3 gpurun_out/r03a_pytest.log
