set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for a in "i8 32 dot 8" "i8 48 dot 64" "i8 48 dot 1024" "i8 32 l2 1024" "i8 20 cos 1024" "i8 100 dot 1024"; do timeout -k 10 300 python3 tools/run_c2.py $a; done > gpurun_out/i8s_k.log 2>&1
grep C2 gpurun_out/i8s_k.log
