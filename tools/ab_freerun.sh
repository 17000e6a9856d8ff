for kv in KX_DA=0 KX_DA=1 KX_DA=0 KX_DA=1; do
  env $kv timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 1 --pcie 0 --serve 0 --reduced 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$kv', 'pinned %.2f ms' % d['ms_per_step'], 'free-run wall %.4f s rtf %.0f' % (d['free_running']['wall_s'], d['free_running']['rtf_rank0']))"
done
for b in 1; do timeout -k 10 200 python bench.py --batch 1 --steps 20 --warmup 3 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('batch 1: %.2f ms/step' % d['ms_per_step'])"; done
