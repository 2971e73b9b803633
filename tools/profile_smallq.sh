#!/bin/bash
# Kernel trace + PMC passes of a small-batch int8 call at the C2 corpus (run through gpurun from the repo root):
#   bash tools/profile_smallq.sh r03 1       -> gpurun_out/r03_smallq1_{kernel_stats,pmc_*}.csv     (one query)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
Q=${2:-1}
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_smallq${Q}_stats -- python3 tools/run_c2.py i8 10 dot $Q > gpurun_out/${R}_smallq${Q}_under_rocprof.txt 2> gpurun_out/${R}_smallq${Q}_stats.err
find gpurun_out/${R}_smallq${Q}_stats -name '*kernel_stats.csv' > gpurun_out/${R}_smallq${Q}_files.txt
while read f; do cp "$f" gpurun_out/${R}_smallq${Q}_kernel_stats.csv; done < gpurun_out/${R}_smallq${Q}_files.txt
echo "kernel trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    name=${pass%% *}
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/${R}_smallq${Q}_pmc_$name -- python3 tools/run_c2.py i8 10 dot $Q > gpurun_out/${R}_smallq${Q}_pmc_$name.log 2>&1 || echo "pass $name failed"
    python3 tools/pmc_summary.py gpurun_out/${R}_smallq${Q}_pmc_$name gpurun_out/${R}_smallq${Q}_pmc_$name.csv || true
    echo "pmc $name done"
done
grep -h "gemm_i8s" gpurun_out/${R}_smallq${Q}_kernel_stats.csv gpurun_out/${R}_smallq${Q}_pmc_*.csv | cut -c1-200
