timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5 && \
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 && \
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var_dual.txt 2>&1 | grep -E "timed" && \
head -12 gpurun_out/var_dual.txt && \
KX_DUAL=0 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 2>&1 | grep -E "timed"
