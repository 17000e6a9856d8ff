timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 && \
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 && \
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var_pair.txt 2>&1 | grep timed && head -10 gpurun_out/var_pair.txt && \
KX_PAIR=0 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 2>&1 | grep timed
