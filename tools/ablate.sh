for cfg in "KX_DBG=9" "KX_DBG=11" "KX_DBG=9 KX_BN=128 KX_TK=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var.txt 2>&1 | grep -E "timed"
  head -6 gpurun_out/var.txt | tail -5
done
