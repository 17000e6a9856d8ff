for i in 1; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var.txt 2>&1 | grep -E "timed"
  head -12 gpurun_out/var.txt; grep -E "^ *(2304|2048|768) +[0-9]+ +1 " gpurun_out/var.txt | head -5
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --batch 1 --cpu-utts 0 --free-run 0 2>&1 | grep -E "timed"
done
