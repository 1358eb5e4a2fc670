#!/bin/bash
# PMC passes for the LDS-resident QP kernel (instruction mix, stall reasons). Output: gpurun_out/pmc_small.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# everything is built BEFORE the first rocprofv3 line (hipcc / make / g++ must never run as children of a profiled,
# GPU-initialised process); the profiled bench.py runs get --no-build
python3 __graft_entry__.py > /dev/null || exit 1
i=0
while read -r c; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$i -- python3 bench.py --no-build --no-extras --no-large --steps 3 --warmup 1 --stat-launches 2 > gpurun_out/pmc_log_$i.txt 2>&1 || echo "fail $i"
done <<'LIST'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT
SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
GRBM_GUI_ACTIVE SQ_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
LIST
python tools/pmc_summary.py /tmp/pmc_[0-9]* > ${1:-gpurun_out/pmc_small.json}
tail -2 gpurun_out/pmc_log_1.txt
