cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for w in 20 19 18 17 16 12; do
  echo "== d_base waves $w"
  SRT_WAVES_PER_CU=$w SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/d_base/libsrt_hip.so timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1
done > gpurun_out/r04/s4_waves.log 2>&1
cat gpurun_out/r04/s4_waves.log
