cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s6_direct.log
: > $OUT
for v in d_base dr0 dr1 dr2 dr0_hq56 dr0_hq64 dr1_hq64 dr0_hq72 d_base; do
  echo "== $v" >> $OUT
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1 >> $OUT
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/sched_probe.py spheres 2>&1 | grep workload | cut -c1-330 >> $OUT
done
cut -c1-400 $OUT
