cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for r in 1 2; do
for v in d_base p_valu p_half p_salu; do
  echo "== $v"
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1
done; done > gpurun_out/r04/s5_regime.log 2>&1
for v in p_valu p_salu; do
  echo "== $v waves 16"
  SRT_WAVES_PER_CU=16 SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1
done >> gpurun_out/r04/s5_regime.log 2>&1
cat gpurun_out/r04/s5_regime.log
