#!/bin/bash
# Regenerate the judged artefacts of a round on the GPU box:  bash tools/round_profiles.sh r01_f
# writes gpurun_out/<tag>_bench.json, <tag>_kernel_stats_bench.csv, <tag>_pmc_hbm_traffic.json,
# <tag>_kernel_stats_blocked_qr.csv  (copy them into profiles/)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# everything is built BEFORE the first rocprofv3 line (hipcc / make / g++ must never run as children of a profiled,
# GPU-initialised process); the profiled bench.py runs get --no-build
python3 __graft_entry__.py > /dev/null || exit 1
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --no-build --no-large --cpu-seconds 1 --steps 20 --warmup 3 --stat-launches 20 > gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
# bench.py starts the C++ host program for the single-QP latency leg: one stats file per process, keep the parent's (largest)
cp "$(find /tmp/prof_bench -name '*kernel_stats.csv' -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2-)" gpurun_out/${tag}_kernel_stats_bench.csv
# the HEADLINE kernel alone: --no-extras launches nothing but the 65 536-QP batch, so the average of this file
# IS the kernel_ms of the bench line (the file above mixes batch sizes: the extras launch the same kernel on 16 384 QPs)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_head -- python3 bench.py --no-build --no-extras --steps 200 --warmup 10 --stat-launches 50 > gpurun_out/${tag}_bench_headline_only.json 2>/dev/null
cp "$(find /tmp/prof_head -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats_headline_only.csv
echo "stats done"
bash tools/pmc_hbm.sh gpurun_out/${tag}_pmc_hbm_traffic.json
echo "pmc hbm done"
# blocked QR on the matrix cores: kernel stats + MFMA / VALU busy counters
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_qr -- python3 tools/qr_profile_run.py > gpurun_out/${tag}_qr_run.txt 2>/dev/null
cp "$(find /tmp/prof_qr -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats_blocked_qr.csv
# MFMA / LDS counters of the GEMM alone (a PMC pass over the ~9000 launches of the QR takes minutes and
# has hung once: not part of the default set) -- see scratch-free recipe in DESIGN.md section 6
echo "qr done"
ls -la gpurun_out | grep ${tag}
