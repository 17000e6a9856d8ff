timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var_up.txt 2>&1 | grep -E "timed"
grep -E " 2$| 2 +[0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+$" gpurun_out/var_up.txt | head; grep -E "^ *(768|2560) " gpurun_out/var_up.txt
