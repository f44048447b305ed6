#!/bin/bash
# usage: scripts/ab_probe.sh <workloads> <variant>...   (GPU box) — sched_probe.py for each lib/variants/<variant>
W=$1; shift
for v in "$@"; do
  echo "== $v"
  SRT_LIB=simple-raytracer_amd/lib/variants/$v/libsrt_hip.so python scripts/sched_probe.py $W 2>&1 | grep workload
done
