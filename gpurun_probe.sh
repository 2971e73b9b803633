set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 500 python3 tools/bench_midq.py > gpurun_out/r03_midq_10Mx768.txt 2> gpurun_out/midq.err
grep -E "AUTO" gpurun_out/r03_midq_10Mx768.txt | cut -c1-140
