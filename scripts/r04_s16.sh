cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_collect.py tests/test_gpu_bvh.py tests/test_host_mirror.py -m gpu -x -q > gpurun_out/r04/s16_pytest.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r04/s16_pytest.log
for n in 1 8 1 8; do timeout -k 10 120 python scripts/bvh_update_probe.py $n 2>&1 | tail -1; done | tee gpurun_out/r04/bvh_update_probe.jsonl
timeout -k 10 120 ./simple-raytracer_amd/lib/srt_headless --scene spheres --width 320 --height 200 --spp 8 --gpus 4 --out gpurun_out/r04/headless_gpus4.ppm 2>&1 | tail -2
timeout -k 10 120 ./simple-raytracer_amd/lib/srt_headless --scene spheres --width 320 --height 200 --spp 8 --out gpurun_out/r04/headless_gpus1.ppm 2>&1 | tail -2
cmp gpurun_out/r04/headless_gpus4.ppm gpurun_out/r04/headless_gpus1.ppm && echo "headless --gpus 4 == --gpus 1"
rm -f gpurun_out/r04/headless_gpus*.ppm
