# SQ counter passes (tools/pmc_sq.sh) summarised for SEVERAL kernels from the same five passes:
#   bash tools/pmc_sq_multi.sh TAG "kernel substring 1" "kernel substring 2" ...
# one rocprofv3 --pmc run per counter group over `bench.py --steps 1 --warmup 0` (no tracing flags beside --pmc).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq_$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 --latency-b1 0 > $R/gpurun_out/${TAG}_pmc_sq_$i.log 2>&1
  echo "pass $i done"
done
cd $R
n=0
for k in "$@"; do
  n=$((n+1))
  python tools/summarize_pmc_sq.py gpurun_out/${TAG}_pmc_sq_ "$k" > gpurun_out/${TAG}_pmc_sq_kernel$n.txt
  head -3 gpurun_out/${TAG}_pmc_sq_kernel$n.txt
done
# keep the merged-back directory small: the raw csvs are tens of MB
rm -rf gpurun_out/${TAG}_pmc_sq_[0-9]
