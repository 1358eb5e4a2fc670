#!/bin/bash
# rocprofv3 kernel statistics of the HBM-resident engine on BASELINE configs 3 and 4:  bash tools/large_profiles.sh r02_x [workloads]
# (workloads: any of dense sparse band5; default "dense sparse band5")
tag=${1:-rXX}
loads=${2:-dense sparse band5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 __graft_entry__.py > /dev/null || exit 1        # build outside the profiler
for w in $loads; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_large_$w -- python3 tools/large_profile_run.py $w 4 > gpurun_out/${tag}_large_${w}_run.txt 2>/dev/null
  cp "$(find /tmp/prof_large_$w -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats_large_${w}.csv
  cat gpurun_out/${tag}_large_${w}_run.txt
done
