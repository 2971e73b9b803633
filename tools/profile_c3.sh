#!/bin/bash
# C3 evidence on a 1-GPU box (run through gpurun from the repo root): batch_knn_u8 50M x 768, 1024 queries, k = 100 on the
# int8-MFMA filter engine -- kernel trace + PMC passes of `python3 tools/bench_u8.py N i8`.
#   1. plain run                                  -> gpurun_out/<R>_c3_i8.json
#   2. rocprofv3 --kernel-trace --stats           -> gpurun_out/<R>_c3_i8_kernel_stats.csv
#   3. separate --pmc passes (never with a trace) -> gpurun_out/<R>_c3_i8_pmc_<first counter>.csv
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r02}
N=${2:-50000000}
mkdir -p gpurun_out
python3 tools/bench_u8.py $N i8 > gpurun_out/${R}_c3_i8.json 2> gpurun_out/${R}_c3_i8.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_c3_stats -- python3 tools/bench_u8.py $N i8 > gpurun_out/${R}_c3_i8_under_rocprof.json 2> gpurun_out/${R}_c3_stats.err
f=$(find gpurun_out/${R}_c3_stats -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${R}_c3_i8_kernel_stats.csv
echo "kernel trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_LEVEL_VMEM"; do  # (a TA_* pass crashed rocprofv3 on this pool: left out)
    name=${pass%% *}
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/${R}_c3_pmc_$name -- python3 tools/bench_u8.py $N i8 > gpurun_out/${R}_c3_pmc_$name.log 2>&1 || echo "pass $name failed"
    python3 tools/pmc_summary.py gpurun_out/${R}_c3_pmc_$name gpurun_out/${R}_c3_i8_pmc_$name.csv || true
    echo "pmc $name done"
done
cat gpurun_out/${R}_c3_i8.json
