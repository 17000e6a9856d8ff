# rocprofv3 kernel stats of the pinned (equal-length) step and of the free-running (ragged) step, same command otherwise,
# and the per-kernel attribution of the difference per frame of audio (tools/ragged_attrib.py).  Run on the GPU box.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
COMMON="--steps 5 --warmup 2 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_pinned -- python3 $R/bench.py $COMMON > $R/gpurun_out/${TAG}_prof_pinned.json 2> $R/gpurun_out/${TAG}_prof_pinned.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_free -- python3 $R/bench.py $COMMON --durations free > $R/gpurun_out/${TAG}_prof_free.json 2> $R/gpurun_out/${TAG}_prof_free.log
cd $R
python tools/ragged_attrib.py gpurun_out/${TAG}_prof_pinned gpurun_out/${TAG}_prof_pinned.json gpurun_out/${TAG}_prof_free gpurun_out/${TAG}_prof_free.json gpurun_out/${TAG}_ragged_attribution.txt
