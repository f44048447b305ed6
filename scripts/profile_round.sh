#!/bin/bash
# usage: scripts/profile_round.sh <tag>   (GPU box, via gpurun) -- every measurement DESIGN.md / profiles/ quote for a round:
# the headline bench line, rocprofv3 kernel stats of the same command, PMC groups at development size (64 spp), HBM
# traffic counters at full size (separate passes), and the other BASELINE configs at full size.
set -u
TAG=$1
R=$PWD
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1 ); echo "stats rc=$?"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
scripts/pmc_cmd.sh ${TAG}_spp64 sq1,sq2,mix,flow,cache,grbm scripts/sched_probe.py spheres > $OUT/pmc_spp64.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_${TAG}_spp64/summary.json $OUT/pmc_spp64.json
( cd /tmp && export TMPDIR=/tmp && for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$c.log 2>&1; echo "$c rc=$?"; done )
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/*_SIZE/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if "srt_" not in k: continue
        a = agg[k + "." + row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
json.dump({k: {"raw_per_launch": s / n, "launches": n} for k, (s, n) in agg.items()}, open("$OUT/traffic_raw.json", "w"), indent=1, sort_keys=True)
print(open("$OUT/traffic_raw.json").read())
PY
python bench.py --workload meshes_1080p_512spp --steps 3 --no-cpu-baseline > $OUT/bench_meshes_scan.json 2>/dev/null; echo "meshes scan rc=$?"
python bench.py --workload meshes_1080p_512spp --accel bvh --steps 3 --no-cpu-baseline > $OUT/bench_meshes_bvh.json 2>/dev/null; echo "meshes bvh rc=$?"
python bench.py --workload mesh100k_1080p_256spp --accel bvh --steps 3 --no-cpu-baseline > $OUT/bench_mesh100k_bvh.json 2>/dev/null; echo "mesh100k bvh rc=$?"
python bench.py --workload spheres_4k_4096spp --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_4k_1gpu.json 2>/dev/null; echo "4k rc=$?"
python bench.py --workload spheres_256_16spp --steps 20 --warmup 2 > $OUT/bench_config0.json 2>/dev/null; echo "config0 rc=$?"
for f in $OUT/bench*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['roofline']['frac'])")"; done
