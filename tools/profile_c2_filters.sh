#!/bin/bash
# Kernel trace + PMC passes of the C2 call on one filter engine (run through gpurun from the repo root):
#   bash tools/profile_c2_filters.sh r03 i8        -> gpurun_out/r03_c2_i8_{kernel_stats,pmc_*}.csv
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
E=${2:-i8}
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_c2_${E}_stats -- python3 tools/run_c2.py $E > gpurun_out/${R}_c2_${E}_under_rocprof.txt 2> gpurun_out/${R}_c2_${E}_stats.err
cp "$(find gpurun_out/${R}_c2_${E}_stats -name '*kernel_stats.csv' | head -1)" gpurun_out/${R}_c2_${E}_kernel_stats.csv
echo "kernel trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_LEVEL_VMEM"; do
    name=${pass%% *}
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/${R}_c2_${E}_pmc_$name -- python3 tools/run_c2.py $E > gpurun_out/${R}_c2_${E}_pmc_$name.log 2>&1 || echo "pass $name failed"
    python3 tools/pmc_summary.py gpurun_out/${R}_c2_${E}_pmc_$name gpurun_out/${R}_c2_${E}_pmc_$name.csv || true
    echo "pmc $name done"
done
