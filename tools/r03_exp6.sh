#!/bin/bash
# round 3, run 6 (one box): input prefetch as inline asm with hand-counted waits -- kernel + forward tests, A/B against the previous build
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_forward.py -x -q -m gpu > gpurun_out/r03h_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r03h_pytest.log
[ $rc -ne 0 ] && exit $rc
tools/ab_variants.sh r03h main w64
