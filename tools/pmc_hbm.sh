#!/bin/bash
# HBM traffic counters (separate passes, MI355X_MICROARCH.md) for the kernels of `bench.py --no-large`.
# Usage (on the GPU box): bash tools/pmc_hbm.sh <out.json>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_hbm_traffic.json}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmch_$c -- python3 bench.py --no-large --cpu-seconds 1 --steps 3 --warmup 1 > gpurun_out/pmch_$c.log 2>&1 || echo "fail $c"
done
python tools/pmc_summary.py /tmp/pmch_FETCH_SIZE /tmp/pmch_WRITE_SIZE > "$out"
