#!/bin/bash
# PMC passes (instruction mix, waits, LDS, instruction cache) for the KKT-tableau kernel (qp_small_g.h) on the 14
# members of 69 x 28 of the hs0xx batch. Output: $1 (default gpurun_out/pmc_k.json)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 __graft_entry__.py > /dev/null || exit 1
i=0
while read -r c; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmck_$i -- python3 tools/k_run.py > gpurun_out/pmck_log_$i.txt 2>&1 || echo "fail $i"
done <<'LIST'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_ANY
SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
LIST
python tools/pmc_summary.py /tmp/pmck_[0-9]* > ${1:-gpurun_out/pmc_k.json}
