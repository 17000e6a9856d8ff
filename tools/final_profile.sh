# Round-end measurement set (run on the GPU box through gpurun): tests, default bench (with CPU baseline and the
# free-running step), f32-mode bench, rocprofv3 kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes).
set -e
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/final_bench_f16x3.json 2> gpurun_out/final_bench_f16x3.log
tail -1 gpurun_out/final_bench_f16x3.json | cut -c1-400
KOKOROX_CONV=f32 python bench.py --cpu-utts 0 > gpurun_out/final_bench_f32.json 2> gpurun_out/final_bench_f32.log
tail -1 gpurun_out/final_bench_f32.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 > $R/gpurun_out/final_prof_bench.json 2> $R/gpurun_out/final_prof.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final_pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-utts 0 --free-run 0 --pcie 0 > $R/gpurun_out/final_pmc_fetch.log 2> $R/gpurun_out/final_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final_pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-utts 0 --free-run 0 --pcie 0 > $R/gpurun_out/final_pmc_write.log 2>&1
cd $R
python tools/summarize_rocprof.py gpurun_out/final_prof gpurun_out/final_prof_bench.json gpurun_out/final_kernel_stats.txt
head -12 gpurun_out/final_kernel_stats.txt | cut -c1-150
python tools/summarize_pmc.py gpurun_out/final_pmc_fetch gpurun_out/final_pmc_write gpurun_out/final_pmc_conv_traffic_f16x3.json gpurun_out/final_pmc_fetch.log "conv1d_f16x3_kernel<128," f16x3
cat gpurun_out/final_pmc_conv_traffic_f16x3.json
