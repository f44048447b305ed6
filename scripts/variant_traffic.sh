#!/bin/bash
# usage: scripts/variant_traffic.sh name "flags" ...   (pairs) -> time + HBM traffic per variant at 64 spp
R=$PWD
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  lib=$(python3 -c "
import sys; sys.path.insert(0,'$R')
import srt_pkg; srt_pkg.load()
from simple_raytracer_amd import build as B
print(B.build_variant('$name', '$flags'.split()))")
  echo "== $name ($flags)"
  SRT_LIB=$lib timeout -k 10 120 python3 scripts/time_one.py spheres64 3
  SRT_LIB=$lib timeout -k 10 200 bash scripts/pmc_traffic.sh $name --spp 64 --steps 1 --warmup 0 --no-cpu-baseline | grep trace
  cd $R
done
