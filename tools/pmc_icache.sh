#!/bin/bash
# instruction-cache counters of the LDS-resident QP kernel. Output: gpurun_out/pmc_icache.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# everything is built BEFORE the first rocprofv3 line (hipcc / make / g++ must never run as children of a profiled,
# GPU-initialised process); the profiled bench.py runs get --no-build
python3 __graft_entry__.py > /dev/null || exit 1
i=0
while read -r c; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmci_$i -- python3 bench.py --no-build --no-extras --no-large --steps 3 --warmup 1 --stat-launches 2 > gpurun_out/pmci_log_$i.txt 2>&1 || echo "fail $i"
done <<'LIST'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_INSTS_BRANCH
SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
LIST
python tools/pmc_summary.py /tmp/pmci_* > gpurun_out/pmc_icache.json
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|inst_fetch" | head -40 > gpurun_out/avail_icache.txt
