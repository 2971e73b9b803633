set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_u8.py -x -q -m gpu -k "small_batch" > gpurun_out/i8s_tests.log 2>&1 || { tail -40 gpurun_out/i8s_tests.log; exit 1; }
tail -2 gpurun_out/i8s_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_i8s -o i8s -- python3 $GRAFT_REPO_ROOT/tools/run_c2.py i8 10 dot 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_i8s.log 2>&1
cd $GRAFT_REPO_ROOT && ls gpurun_out/prof_i8s | head; f=$(ls gpurun_out/prof_i8s/*kernel_stats.csv | head -1); head -25 $f | cut -c1-160
