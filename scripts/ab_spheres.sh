#!/bin/bash
# usage: scripts/ab_spheres.sh name...   (GPU box): trace-kernel time of BASELINE configs[1] for lib/variants/<name> ("main" = the shipped library), interleaved, two rounds
set -u
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset SRT_LIB; else export SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so; fi
    timeout -k 10 120 python scripts/time_full.py spheres 3 > gpurun_out/ab_sph_one.json || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/ab_sph_one.json'))
print('$v'.ljust(12), ' '.join(f\"{k} {v['kernel_ms_min']:.2f} ms {v['kernel_ms_all']} (checksum {v['checksum']:.6f})\" for k, v in d.items()))"
  done
done
