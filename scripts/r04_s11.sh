cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/s11_pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04/s11_pytest.log
timeout -k 10 300 python scripts/bench_stall_probe.py > gpurun_out/r04/bench_stall_probe.jsonl 2> gpurun_out/r04/bench_stall_probe.err; echo "probe rc $?"; cat gpurun_out/r04/bench_stall_probe.jsonl
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > gpurun_out/r04/s11_bench.json 2> gpurun_out/r04/s11_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r04/s11_bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04/s11_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','kernel_ms')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_equals_port'])
for o in d['other_configs']: print({k:o.get(k) for k in ('workload','ms_per_step','ms_per_step_max','kernel_ms','frac','warmup_steps','watchdog','canvas_equals_bvh_canvas','slowest_step_host_ms','error')})
PY
