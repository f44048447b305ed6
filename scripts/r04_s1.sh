set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/s1_pytest.log 2>&1; echo "pytest rc $?"
for v in d_base d_hq44 d_hq48 d_hq52 d_hq56r40 d_hq64r32 d_hq64 d_hq80; do
  echo "== $v"
  SRT_LIB=simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/sched_probe.py spheres 2>&1 | grep workload
done > gpurun_out/r04/s1_hq.log 2>&1
cat gpurun_out/r04/s1_hq.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/r04/s1_bench.json 2> gpurun_out/r04/s1_bench.err; echo "bench rc $?"
tail -c 3000 gpurun_out/r04/s1_bench.json
