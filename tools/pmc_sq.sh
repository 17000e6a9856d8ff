# SQ counter passes for the dominant conv kernel (one rocprofv3 --pmc run per counter group; no tracing flags)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_sq_$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 > $R/gpurun_out/pmc_sq_$i.log 2>&1
  echo "pass $i done"
done
cd $R
python tools/summarize_pmc_sq.py gpurun_out/pmc_sq_ "${1:-conv1d_f16x3_da_kernel<2, 11, 6, false, false, true>}"
