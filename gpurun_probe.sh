set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_u8.py tests/test_gpu_mfma.py -x -q -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1 || { tail -30 gpurun_out/profile_round.log; exit 1; }
tail -1 gpurun_out/profile_round.log | cut -c1-300
