#!/bin/bash
# Round-3 additions to tools/round_profiles.sh (run on the GPU box):  bash tools/round3_profiles.sh r03_a
#   <tag>_stamps_kkt_inverse_69x28.txt      cycles per phase of the KKT-tableau kernel (block 0)
#   <tag>_pmc_kkt_inverse.json              instruction mix / waits / LDS / instruction cache of that kernel
#   <tag>_pmc_mfma_blocked_setup.json       SQ_VALU_MFMA_BUSY_CYCLES & co. of the blocked QR + Q + R^-1 (a 4096 x 3072 instance:
#                                           a counter pass serialises every one of the launches; the 10 000 x 7 670 one takes minutes)
tag=${1:-r03_x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 __graft_entry__.py > /dev/null || exit 1
python3 tools/stamp_k_kernel.py > gpurun_out/${tag}_stamps_kkt_inverse_69x28.txt 2>&1
echo "stamps done"
bash tools/pmc_k.sh gpurun_out/${tag}_pmc_kkt_inverse.json
echo "pmc k done"
i=0
while read -r c; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcm_$i -- python3 tools/qr_profile_run.py 4096 3072 > gpurun_out/pmcm_log_$i.txt 2>&1 || echo "fail $i"
done <<'LIST'
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA
SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
LIST
python tools/pmc_summary.py /tmp/pmcm_[0-9]* > gpurun_out/${tag}_pmc_mfma_blocked_setup.json
echo "pmc mfma done"
ls -la gpurun_out | grep ${tag}
