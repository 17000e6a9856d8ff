for cfg in "KX_MERGE=1" "KX_MERGE=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var.txt 2>&1 | grep -E "timed"
  grep -E "^ *(2304|2048|768) " gpurun_out/var.txt
done
