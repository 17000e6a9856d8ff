set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1b -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 > $R/gpurun_out/prof_r1b_bench.json 2> $R/gpurun_out/prof_r1b.log
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_r1b gpurun_out/prof_r1b_bench.json gpurun_out/kernel_stats_new.txt
head -32 gpurun_out/kernel_stats_new.txt | cut -c1-140
