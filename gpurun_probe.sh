set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for a in "i8 100 dot 8" "i8 100 dot 64" "i8 100 l2 64" "i8 100 dot 128" "i8 100 dot 1" "i8 100 dot 1024" "i8 200 dot 64"; do timeout -k 10 300 python3 tools/run_c2.py $a; done > gpurun_out/i8s_k.log 2>&1
grep C2 gpurun_out/i8s_k.log
