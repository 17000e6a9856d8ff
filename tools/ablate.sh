for d in 0 1 2 4 8 7 15; do
  echo "== KX_DBG=$d"
  KX_DBG=$d timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/abl_$d.txt 2>&1 | grep -E "timed"
  head -4 gpurun_out/abl_$d.txt | tail -3
done
