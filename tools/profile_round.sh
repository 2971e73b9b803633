#!/bin/bash
# Collect the round's judged evidence on a 1-GPU box (run through gpurun from the repo root):
#   1. bench.py (N=1 default) -> gpurun_out/bench_n1.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/prof_stats/
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) -> gpurun_out/pmc_*/ + summaries
# Afterwards copy the summaries into profiles/ (tools/pmc_summary.py writes them in the committed format).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
echo "bench done"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --no-cpu-baseline > gpurun_out/${R}_bench_n1_under_rocprof.json 2> gpurun_out/prof_stats.err
echo "kernel trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    name=${pass%% *}
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1
    python3 tools/pmc_summary.py gpurun_out/pmc_$name gpurun_out/${R}_bench_n1_pmc_$name.csv
    echo "pmc $name done"
done
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${R}_bench_n1_kernel_stats.csv
cat gpurun_out/bench_n1.json
