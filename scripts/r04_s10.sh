cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s10.log
: > $OUT
for v in diag m_hq62 m_hq60 m_hq56 m_hq52 m_hq48 diag; do
  echo "== $v" >> $OUT
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1 | cut -c1-100 >> $OUT
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/sched_probe.py spheres 2>&1 | grep workload | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel_ms','iterations','shade_phases','waves_per_cu','rays_per_iteration')})" >> $OUT
done
cat $OUT
