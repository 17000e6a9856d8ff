timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -3 && \
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --batch 1 --cpu-utts 0 --free-run 0 2>&1 | grep timed && \
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 2>&1 | grep timed
