cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "draws_the_scene" 2>&1 | tail -3
for w in spheres mesh2 mesh100k_bvh; do
  spp=64; [ $w = mesh2 ] && spp=128
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/regions/libsrt_hip.so timeout -k 10 200 python scripts/region_probe.py $w $spp > gpurun_out/r04/region_freq_${w}_spp${spp}.json 2>> gpurun_out/r04/s25.err; echo "$w rc $?"
done
timeout -k 10 300 python scripts/interactive_probe.py > gpurun_out/r04/interactive.jsonl 2>> gpurun_out/r04/s25.err; cat gpurun_out/r04/interactive.jsonl
