for cfg in "KX_BN=0 KX_TK=3" "KX_BN=192 KX_TK=2" "KX_BN=192 KX_TK=2 KX_PF=0" "KX_BN=192 KX_TK=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/var.txt 2>&1 | grep -E "timed"
  head -6 gpurun_out/var.txt | tail -5
done
