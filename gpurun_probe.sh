set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 500 python3 tools/bench_midq.py > gpurun_out/r03_midq_10Mx768.txt 2> gpurun_out/midq.err
grep -E "AUTO" gpurun_out/r03_midq_10Mx768.txt | cut -c1-120
for a in "i8 100 dot 256" "i8 10 l2 256" "i8 10 cos 200"; do timeout -k 10 300 python3 tools/run_c2.py $a; done > gpurun_out/i8s_k.log 2>&1
grep C2 gpurun_out/i8s_k.log
