for cfg in "KX_W8=1" "KX_W8=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var.txt 2>&1 | grep -E "timed"
  head -7 gpurun_out/var.txt | tail -6
done
