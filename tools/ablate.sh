# A/B harness used for the kernel experiments recorded in DESIGN.md: run on the GPU box through gpurun, e.g.
#   gpurun -- 'KX_DBG=4 bash tools/ablate.sh'      (KX_DBG bits: 1 no input staging, 2 no weight copies, 4 no MFMAs,
#                                                   8 no epilogue; KX_LDS_PAD=30000 forces one workgroup per CU)
# Prints the step time and the per-shape table of the conv launches of one step.
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --free-run 0 --detail gpurun_out/ablate_shapes.txt 2>&1 | grep -E "timed"
head -14 gpurun_out/ablate_shapes.txt
