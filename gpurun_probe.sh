set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 400 python3 tools/bench_u8.py > gpurun_out/r03_c3_i8.json 2> gpurun_out/c3.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03_c3_i8.json')); print('C3', d['total_ms'], d['gemm_ms'])"
