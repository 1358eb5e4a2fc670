#!/bin/bash
# PMC passes for the LDS-resident QP kernel (instruction mix, stall reasons). Output: gpurun_out/pmc_small.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
while read -r c; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$i -- python3 bench.py --no-extras --no-large --steps 3 --warmup 1 > gpurun_out/pmc_log_$i.txt 2>&1 || echo "fail $i"
done <<'LIST'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT
LIST
python tools/pmc_summary.py /tmp/pmc_* > gpurun_out/pmc_small.json
tail -2 gpurun_out/pmc_log_1.txt
