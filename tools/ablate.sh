timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
KOKOROX_CONV=f32 python bench.py --cpu-utts 0 > gpurun_out/final_bench_f32.json 2> gpurun_out/final_bench_f32.log
tail -1 gpurun_out/final_bench_f32.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
KOKOROX_CONV=f32 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 > $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32.log
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/final_prof_f32 gpurun_out/final_prof_f32_bench.json gpurun_out/final_kernel_stats_f32.txt
head -10 gpurun_out/final_kernel_stats_f32.txt | cut -c1-150
