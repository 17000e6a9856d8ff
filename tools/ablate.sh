./tools/probes/sin_accuracy.bin
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var_sin.txt 2>&1 | grep -E "timed"
head -9 gpurun_out/var_sin.txt
