# shapes outside the bench default, run on the GPU box: large batch, 500-phoneme chunks (T = 502), tiny inputs
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --batch 256 --cpu-utts 0 --free-run 0 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --batch 16 --phonemes 500 --cpu-utts 0 --free-run 0 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --batch 2 --phonemes 3 --cpu-utts 0 --free-run 0 2>&1 | tail -1 | cut -c1-200
