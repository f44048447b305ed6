#!/bin/bash
# usage: scripts/bvh_soak_round.sh <tag> <n> <seed>   (GPU box) -- tests/bvh_soak.py: the BVH against the array scan on procedural meshes of 100 .. 6,000 triangles
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1_bvh_soak.txt
echo "# tests/bvh_soak.py on one MI355X, kernels of $(sha256sum simple-raytracer_amd/csrc/kernels.hip | cut -c1-12): python tests/bvh_soak.py $2 $3" > $OUT
timeout -k 10 1100 python tests/bvh_soak.py $2 $3 >> $OUT 2>&1   # (progress lines go straight to the file: a silent run is taken for hung)
tail -3 $OUT
