#!/bin/bash
# usage: scripts/fuzz_soak_round.sh <tag> <seed>   (GPU box) -- the differential fuzz soak of a round's final kernels, array scan and BVH
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1_fuzz_soak.txt
echo "# tests/fuzz_soak.py on one MI355X, kernels of $(sha256sum simple-raytracer_amd/csrc/kernels.hip | cut -c1-12) (differential: GPU canvas and counters against the CPU oracle, 24x16 canvases)" > $OUT
echo "## array scan (parity mode): python tests/fuzz_soak.py 120000 120000 $2" >> $OUT
timeout -k 10 1000 python tests/fuzz_soak.py 120000 120000 $2 > $OUT.a 2>&1; tail -2 $OUT.a >> $OUT   # (progress goes to a file: a silent run is taken for hung)
echo "## BVH: SRT_FUZZ_ACCEL=1 python tests/fuzz_soak.py 120000 120000 $(( $2 + 1 ))" >> $OUT
SRT_FUZZ_ACCEL=1 timeout -k 10 1000 python tests/fuzz_soak.py 120000 120000 $(( $2 + 1 )) > $OUT.b 2>&1; tail -2 $OUT.b >> $OUT
cat $OUT
