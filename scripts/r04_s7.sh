cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/s7_pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04/s7_pytest.log
OUT=gpurun_out/r04/s7_full.log
: > $OUT
timeout -k 10 300 python scripts/time_full.py spheres,meshes,meshes:bvh,mesh100k:bvh 2 2>&1 | tail -1 >> $OUT
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/diag/libsrt_hip.so timeout -k 10 120 python scripts/sched_probe.py spheres,mesh2,mesh2_bvh,mesh100k_bvh 2>&1 | grep workload | cut -c1-400 >> $OUT
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/regions/libsrt_hip.so timeout -k 10 120 python scripts/region_probe.py spheres > gpurun_out/r04/region_freq_spheres_spp64_a.json 2>> $OUT
cat $OUT
