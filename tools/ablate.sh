for d in 0; do
  echo "== KX_DBG=$d"
  KX_DBG=$d timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var_$d.txt 2>&1 | grep -E "timed"
  grep -E "^ *(128|256) +(128|256) +(11|3) +1 " gpurun_out/var_$d.txt | head -4
done
