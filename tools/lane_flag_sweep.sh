#!/bin/bash
# Tuning: compiler-flag variants of the lane-per-problem kernel (qp_lane.hip); launch time of the headline batch each.
# Run on the GPU box:  bash tools/lane_flag_sweep.sh
cd "$(dirname "$0")/.."
i=0
while read -r flags; do
  i=$((i+1))
  echo "== variant $i: $flags"
  if tools/lane_experiment.sh $flags > /dev/null 2>gpurun_out/lane_flag_$i.err; then
    RSQP_LIB=$PWD/restartsqp_amd/lib/librsqp_exp.so timeout -k 10 120 python tools/lane_vs_tiny_sweep.py 65536 2>&1 | tail -1
  else
    echo "build failed"; tail -2 gpurun_out/lane_flag_$i.err
  fi
done <<'LIST'
-DBASE
-mllvm -enable-misched=false
-mllvm -amdgpu-schedule-metric-bias=100
-mllvm -enable-post-misched=false
-mllvm -amdgpu-schedule-relaxed-occupancy=true
-O2
-mllvm -amdgpu-spill-vgpr-to-agpr=false
-mllvm -amdgpu-use-amdgpu-trackers=1
-mllvm -misched-cluster=false
-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=true
LIST
