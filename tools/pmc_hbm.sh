#!/bin/bash
# HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes, MI355X_MICROARCH.md) for the kernels of bench.py.
# Two runs per counter: the timed batch alone (--no-extras; its per-dispatch mean is the figure bench.py's
# roofline.traffic uses) and the extras (SpMV roofline ...). The extras also launch the batch kernel on
# single QPs (latency leg), which would drag that kernel's mean down -- the first run's entry wins.
# Usage (on the GPU box): bash tools/pmc_hbm.sh <out.json>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# everything is built BEFORE the first rocprofv3 line (hipcc / make / g++ must never run as children of a profiled,
# GPU-initialised process); the profiled bench.py runs get --no-build
python3 __graft_entry__.py > /dev/null || exit 1
out=${1:-gpurun_out/pmc_hbm_traffic.json}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcA_$c -- python3 bench.py --no-build --no-extras --no-large --steps 3 --warmup 1 --stat-launches 2 > gpurun_out/pmcA_$c.log 2>&1 || echo "fail A $c"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcB_$c -- python3 bench.py --no-build --no-large --cpu-seconds 1 --steps 3 --warmup 1 --stat-launches 2 > gpurun_out/pmcB_$c.log 2>&1 || echo "fail B $c"
done
python tools/pmc_summary.py /tmp/pmcB_FETCH_SIZE /tmp/pmcB_WRITE_SIZE > /tmp/pmcB.json
python tools/pmc_summary.py /tmp/pmcA_FETCH_SIZE /tmp/pmcA_WRITE_SIZE > /tmp/pmcA.json
python3 - "$out" <<'PY'
import json, sys
b = json.load(open("/tmp/pmcB.json")); a = json.load(open("/tmp/pmcA.json"))
b.update(a)
json.dump(b, open(sys.argv[1], "w"), indent=1, sort_keys=True)
PY
