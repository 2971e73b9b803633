set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 400 python3 tools/bench_maxsim.py > gpurun_out/ms_c4.json 2> gpurun_out/ms_c4.err
python3 -c "
import json; d=json.load(open('gpurun_out/ms_c4.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('scan_ms','total_ms','scan_GBps')})
"
