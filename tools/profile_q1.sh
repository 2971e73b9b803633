#!/bin/bash
# The HBM-bound single-query scans under rocprofv3 (run through gpurun from the repo root): kernel trace + FETCH_SIZE / WRITE_SIZE
# passes of tools/bench_q1.py, summaries -> gpurun_out/<R>_q1_<which>_*.  Achieved HBM GB/s = (FETCH_SIZE x 2 (gfx950 unit
# correction, MI355X_MICROARCH.md) + WRITE_SIZE) KB per dispatch / the kernel's average duration.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r02}
mkdir -p gpurun_out
for which in f32 u8 maxsim; do
    python3 tools/bench_q1.py $which > gpurun_out/${R}_q1_$which.txt 2> gpurun_out/${R}_q1_$which.err
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_q1_${which}_stats -- python3 tools/bench_q1.py $which > /dev/null 2> gpurun_out/${R}_q1_${which}_stats.err || echo "trace $which failed"
    f=$(find gpurun_out/${R}_q1_${which}_stats -name "*kernel_stats.csv" | head -1)
    cp "$f" gpurun_out/${R}_q1_${which}_kernel_stats.csv
    for pass in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/${R}_q1_${which}_pmc_$pass -- python3 tools/bench_q1.py $which > /dev/null 2>&1 || echo "pass $pass failed"
        python3 tools/pmc_summary.py gpurun_out/${R}_q1_${which}_pmc_$pass gpurun_out/${R}_q1_${which}_pmc_$pass.csv || true
    done
    echo "$which done"
done
cat gpurun_out/${R}_q1_*.txt
