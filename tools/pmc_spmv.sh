#!/bin/bash
# PMC passes (instruction mix, waits, LDS, vector memory) for the LDS-vector SpMV kernels on the BASELINE sparse shape.
# Usage: tools/pmc_spmv.sh [out.json] [variants...]   (default gpurun_out/pmc_spmv.json, variant 40)
out=${1:-gpurun_out/pmc_spmv.json}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 __graft_entry__.py > /dev/null || exit 1     # build BEFORE the first rocprofv3 line (no compiler under a profiled process)
i=0
while read -r c; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$i -- python3 tools/spmv_bound_check.py ${@:-40} > gpurun_out/pmcs_log_$i.txt 2>&1 || echo "fail $i"
done <<'LIST'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY GRBM_GUI_ACTIVE
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_INSTS_SMEM
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum
LIST
python tools/pmc_summary.py /tmp/pmcs_[0-9]* > $out
