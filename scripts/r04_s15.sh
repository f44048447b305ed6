cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04/s15_bench.json 2> gpurun_out/r04/s15_bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04/s15_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')})
for o in d['other_configs']: print({k:o.get(k) for k in ('workload','ms_per_step','kernel_ms','launches_per_step','error')})
PY
