# PMC passes over the vocoder at the benchmark shape: per-kernel SQ counters of the streaming ResBlock kernels (and everything else)
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/voc_bench.py"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM"; do
  rm -rf /tmp/pm; timeout 600 rocprofv3 --pmc $set -d /tmp/pm -o pm -- $CMD > /tmp/pm.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/pm -name "*.db" | head -1) --pmc 2>&1 | grep "resstream\|pairstream128_kernel<3\|launches"
done
