cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for r in 1 2; do
for v in d_base d_hq44 d_hq48 d_hq52 d_hq64 d_hq80; do
  echo "== $v"
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1
done; done > gpurun_out/r04/s3_full.log 2>&1
cat gpurun_out/r04/s3_full.log
