#!/bin/bash
# MFMA-busy counters of the matrix-core set-up of the range-space paths (Gram matrix + blocked Cholesky + triangular inverse +
# U^-1 U^-T), per kernel:  bash tools/pmc_mfma_setup.sh r05_x   ->  gpurun_out/<tag>_pmc_mfma_setup.json
# (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes; the build runs BEFORE the first rocprofv3 line)
tag=${1:-r05_x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 __graft_entry__.py > /dev/null || exit 1
python3 tools/setup_profile_run.py > gpurun_out/${tag}_setup_run.txt 2>&1
cat gpurun_out/${tag}_setup_run.txt
i=0
while read -r c; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$i -- python3 tools/setup_profile_run.py > gpurun_out/pmcs_log_$i.txt 2>&1 || echo "fail $i"
done <<'LIST'
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA
SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
LIST
python3 tools/pmc_summary.py /tmp/pmcs_[0-9]* > gpurun_out/${tag}_pmc_mfma_setup.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmcs_stats -- python3 tools/setup_profile_run.py > /dev/null 2>&1
cp "$(find /tmp/pmcs_stats -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats_setup.csv
ls -la gpurun_out | grep ${tag}
