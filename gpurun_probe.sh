set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_u8.py tests/test_gpu_matryoshka.py tests/test_gpu_shards.py -x -q -m gpu > gpurun_out/u8_tests.log 2>&1 || { tail -40 gpurun_out/u8_tests.log; exit 1; }
tail -2 gpurun_out/u8_tests.log
timeout -k 10 800 python3 tools/bench_u8_smallq.py 50000000 100 > gpurun_out/r03_u8_smallq_k100_50Mx768.txt 2> gpurun_out/u8_smallq.err
grep -E "AUTO|#" gpurun_out/r03_u8_smallq_k100_50Mx768.txt
