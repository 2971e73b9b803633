set -e
mkdir -p gpurun_out
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1 || { tail -30 gpurun_out/profile_round.log; exit 1; }
tail -3 gpurun_out/profile_round.log | cut -c1-600
timeout -k 10 500 python3 tools/bench_midq.py > gpurun_out/r03_midq_10Mx768.txt 2> gpurun_out/midq.err
grep -E "AUTO|int8" gpurun_out/r03_midq_10Mx768.txt | cut -c1-140
