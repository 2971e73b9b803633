set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
python3 bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
python3 -c "
import json; d=json.load(open('gpurun_out/bench_n1.json')); print(d['ms_per_step'], d['roofline']['frac'], d['int8_filter_engine']['ms_per_step'], d['bf16_filter_engine']['ms_per_step'], d['lcg_side_row']['ms_per_step'])"
