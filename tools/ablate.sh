for cfg in "KX_BN=0 KX_TK=3" "KX_BN=128 KX_TK=1" "KX_BN=128 KX_TK=2" "KX_BN=0 KX_TK=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var.txt 2>&1 | grep -E "timed"
  head -5 gpurun_out/var.txt | tail -4
done
