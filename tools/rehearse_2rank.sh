# two ranks on ONE GPU (rehearsal of the multi-GPU bench path; the real N>1 runs are the driver's)
KX_SHARE_GPU=1 KX_DIST_BACKEND=${KX_DIST_BACKEND:-gloo} timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 16 --cpu-utts 0 --free-run 0 > gpurun_out/rehearse2.log 2>&1
tail -30 gpurun_out/rehearse2.log | cut -c1-300
