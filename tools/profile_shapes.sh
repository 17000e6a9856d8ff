# rocprofv3 kernel stats of the two side configurations (batch 1, and batch 16 x 500-phoneme chunks)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -- python3 $R/bench.py --steps 10 --warmup 2 --batch 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > $R/gpurun_out/prof_b1_bench.json 2> $R/gpurun_out/prof_b1.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_long -- python3 $R/bench.py --steps 3 --warmup 1 --batch 16 --phonemes 500 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > $R/gpurun_out/prof_long_bench.json 2> $R/gpurun_out/prof_long.log
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_b1 gpurun_out/prof_b1_bench.json gpurun_out/kernel_stats_b1.txt
python tools/summarize_rocprof.py gpurun_out/prof_long gpurun_out/prof_long_bench.json gpurun_out/kernel_stats_long.txt
head -12 gpurun_out/kernel_stats_b1.txt | cut -c1-140
head -12 gpurun_out/kernel_stats_long.txt | cut -c1-140
