#!/bin/bash
# usage: scripts/pmc_traffic.sh <tag> <bench args...>  -> FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
set -u
TAG=$1; shift
R=$PWD; OUT=$R/gpurun_out/traffic_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py "$@" > $OUT/$c.log 2>&1; echo "$c rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        if "srt_" not in k: continue
        agg[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k, row["Counter_Name"])][1] += 1
for (k, c), (s, n) in sorted(agg.items()):
    print(f"{k:42s} {c:11s} {s/n*1024/1e9:9.3f} GB per launch ({n} launches)")
PY
