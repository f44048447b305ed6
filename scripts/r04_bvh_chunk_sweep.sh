#!/bin/bash
export SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/dev/libsrt_hip.so
for c in 6 8 10 12 16 24 32; do
  SRT_JOB_CAP_SUBS=$c timeout -k 10 120 python scripts/time_full.py meshes:bvh,mesh100k:bvh 2 > gpurun_out/sw.json || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/sw.json'))
print('cap $c', ' '.join(f\"{k} {v['kernel_ms_min']:.2f}\" for k, v in d.items()))"
done
