#!/bin/bash
# round 3, run 5 (one box): BN + 64 staged window (split last block) -- kernel + forward tests, then A/B against the BN + 128 build
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_forward.py -x -q -m gpu > gpurun_out/r03f_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r03f_pytest.log
[ $rc -ne 0 ] && exit $rc
tools/ab_variants.sh r03f main w128
