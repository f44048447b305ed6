cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for v in d_base d_hq48 d_hq64; do
  export SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so
  timeout -k 10 300 bash scripts/pmc_cmd.sh r04_$v sq1,sq2 scripts/sched_probe.py spheres > gpurun_out/r04/s2_pmc_$v.log 2>&1
  echo "$v rc $?"
done
