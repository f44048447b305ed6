#!/bin/bash
# usage: scripts/ab_bvh_full.sh <variant>...   (GPU box) -- the two BVH bench lines at full size for each lib/variants/<variant> ("-" = the tree's library)
for v in "$@"; do
  if [ "$v" = "-" ]; then unset SRT_LIB; else export SRT_LIB=simple-raytracer_amd/lib/variants/$v/libsrt_hip.so; fi
  for w in meshes_1080p_512spp mesh100k_1080p_256spp; do
    python bench.py --workload $w --accel bvh --steps 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$w', d['ms_per_step'], 'ms')"
  done
done
