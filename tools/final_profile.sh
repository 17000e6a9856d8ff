# Round-end measurement set (run on the GPU box through gpurun): default bench (with CPU baseline and the free-running
# step), f32-mode bench, rocprofv3 kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes, no tracing flags beside them), batch-1 and long-chunk kernel stats.
#   tools/final_profile.sh TAG [MODE]     MODE = f16f8 (the default mode) or f16x3: the KOKOROX_CONV of every run but the f32 one
set -e
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-final}
MODE=${2:-f16f8}
export KOKOROX_CONV=$MODE
python bench.py > gpurun_out/${TAG}_bench_${MODE}.json 2> gpurun_out/${TAG}_bench_${MODE}.log
tail -1 gpurun_out/${TAG}_bench_${MODE}.json | cut -c1-300
python bench.py --steps 5 --warmup 2 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 --detail gpurun_out/${TAG}_conv_per_shape_${MODE}.txt > gpurun_out/${TAG}_bench_detail.json 2> gpurun_out/${TAG}_bench_detail.log
KOKOROX_CONV=f32 python bench.py --cpu-utts 0 > gpurun_out/${TAG}_bench_f32.json 2> gpurun_out/${TAG}_bench_f32.log
tail -1 gpurun_out/${TAG}_bench_f32.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_b1 -- python3 $R/bench.py --batch 1 --steps 10 --warmup 2 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 > $R/gpurun_out/${TAG}_prof_b1_bench.json 2> $R/gpurun_out/${TAG}_prof_b1.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 > $R/gpurun_out/${TAG}_pmc_fetch.log 2> $R/gpurun_out/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
cd $R
python tools/summarize_rocprof.py gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof_bench.json gpurun_out/${TAG}_kernel_stats_${MODE}.txt > /dev/null
head -12 gpurun_out/${TAG}_kernel_stats_${MODE}.txt | cut -c1-150
python tools/summarize_rocprof.py gpurun_out/${TAG}_prof_b1 gpurun_out/${TAG}_prof_b1_bench.json gpurun_out/${TAG}_kernel_stats_${MODE}_batch1.txt "python3 bench.py --batch 1 --steps 10 --warmup 2 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0" > /dev/null
python tools/summarize_pmc.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_conv_traffic_${MODE}.json gpurun_out/${TAG}_pmc_fetch.log "conv1d_f16x3_kernel<128,|conv1d_f16x3_da_kernel<|conv1d_f16x3_dag_kernel<" $MODE gpurun_out/${TAG}_conv_per_shape_${MODE}.txt
