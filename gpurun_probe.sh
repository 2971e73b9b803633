set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py tests/test_gpu_bf16.py -x -q -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q1 -- python3 tools/run_c2.py i8 10 dot 1 > gpurun_out/prof_q1.log 2> gpurun_out/prof_q1.err
find gpurun_out/prof_q1 -name "*kernel_stats.csv" > gpurun_out/prof_q1_files.txt
while read f; do grep -E "rescore_kernel|query_norms|scan_filter" "$f" | cut -d, -f1-4 | cut -c1-60,100-; done < gpurun_out/prof_q1_files.txt
for a in "i8 10 dot 1" "i8 10 dot 64" "i8 10 dot 1024"; do timeout -k 10 300 python3 tools/run_c2.py $a; done > gpurun_out/i8s_k.log 2>&1
grep C2 gpurun_out/i8s_k.log
