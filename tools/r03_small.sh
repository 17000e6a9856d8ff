#!/bin/bash
# small kernels after a change: forward parity, then rocprof kernel stats of one short bench run (rows of interest printed)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_kernels.py -x -q -m gpu -k "forward or stft or istft or oracle or batch or head" > gpurun_out/r03_small_pytest.log 2>&1 || { tail -20 gpurun_out/r03_small_pytest.log; exit 1; }
tail -2 gpurun_out/r03_small_pytest.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_small_prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > $R/gpurun_out/r03_small_bench.json 2> $R/gpurun_out/r03_small_prof.log
cd $R
python tools/summarize_rocprof.py gpurun_out/r03_small_prof gpurun_out/r03_small_bench.json gpurun_out/r03_small_stats.txt > /dev/null
grep -E "istft|source_|in_stats|pool_up2|style_fc|layernorm|attention|stats_finalize|total GPU" gpurun_out/r03_small_stats.txt | cut -c1-150
python tools/print_bench.py < gpurun_out/r03_small_bench.json
