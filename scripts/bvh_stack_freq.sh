#!/bin/bash
# usage: scripts/bvh_stack_freq.sh   (GPU box, lib/variants/regions built): how often a wave-step of the BVH walk executes its stack instructions
mkdir -p gpurun_out/r04
for w in "mesh100k_bvh 64" "mesh2_bvh 128"; do
  set -- $w
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/regions/libsrt_hip.so timeout -k 10 200 python scripts/region_probe.py $1 $2 > gpurun_out/r04/region_freq_${1}_spp$2.json || exit 1
  python3 - <<PY
import json
d = json.load(open("gpurun_out/r04/region_freq_${1}_spp$2.json")); r = d["regions"]; s = r["EXTEND_BVH_STEP"][0]
print("$1 $2 spp: wave-steps", s, "lanes per step", round(r["EXTEND_BVH_STEP"][1] / s, 1))
for k in r:
    if "BVH_" in k and k != "EXTEND_BVH_STEP": print(" ", k, "per step", round(r[k][0] / s, 3), "lanes", round(r[k][1] / max(r[k][0], 1), 1))
PY
done
