#!/bin/bash
# PMC passes over representative launches (tools/prof_ops.py): MFMA busy, instruction mix, LDS conflicts, waits.
#   tools/pmc_ops.sh <out.txt>      (kernel-trace + --pmc only; one counter group per pass)
set -u
OUT=${1:-gpurun_out/pmc_ops.txt}
export TMPDIR=/tmp
rm -rf /tmp/pmcops; mkdir -p /tmp/pmcops "$(dirname "$OUT")"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
    i=$((i + 1))
    L2D_IGEMM_VARIANT=${L2D_IGEMM_VARIANT:-1} L2D_PROF_REPS=4 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcops/p$i -o p -- \
        python tools/prof_ops.py > /dev/null 2> /tmp/pmcops/err$i.log || tail -3 /tmp/pmcops/err$i.log
done
python tools/pmc_summary.py "$OUT" $(find /tmp/pmcops -name "*.db")
