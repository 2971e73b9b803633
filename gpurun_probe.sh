set -e
mkdir -p gpurun_out
for a in "i8 10 l2" "i8 100 l2" "i8 10 l2 1024 640" "i8 10 l2 64" "i8 10 l2 8" "bf16 10 l2 64"; do timeout -k 10 300 python3 tools/run_c2.py $a; done > gpurun_out/l2_c2.log 2>&1
grep C2 gpurun_out/l2_c2.log
