#!/bin/bash
# usage: scripts/profile_round.sh <tag>   (GPU box, via gpurun) -- every measurement DESIGN.md / profiles/ quote for a round:
# the headline bench line (with the other BASELINE configs at full size in its other_configs), rocprofv3 kernel stats of the same command, PMC groups at development size (64 spp), HBM
# traffic counters at full size (separate passes) -> traffic.json stamped with the kernel source's hash.
set -u
TAG=$1
R=$PWD
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/$c.log 2>&1; echo "$c rc=$?"; done )
python3 - <<PY
import csv, glob, collections, json, hashlib
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/*_SIZE/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if "srt_" not in k: continue
        a = agg[k + "." + row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
raw = {k: {"raw_per_launch": s / n, "launches": n} for k, (s, n) in agg.items()}
json.dump(raw, open("$OUT/traffic_raw.json", "w"), indent=1, sort_keys=True)
# profiles/traffic.json as bench.py reads it: bytes = counter x 1024, per launch of each kernel, stamped with the kernel source's hash
detail = {k: int(v["raw_per_launch"] * 1024) for k, v in raw.items()}
tj = {"kernels_hip_sha256": hashlib.sha256(open("$R/simple-raytracer_amd/csrc/kernels.hip", "rb").read()).hexdigest(),
      "spheres_1080p_1024spp": {"kernel": "srt_trace_kernel", "hbm_bytes_per_launch": detail.get("srt_trace_kernel.FETCH_SIZE", 0) + detail.get("srt_trace_kernel.WRITE_SIZE", 0),
                                "detail_bytes_per_launch": detail,
                                "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (scripts/profile_round.sh), counter values x 1024 B"}}
json.dump(tj, open("$OUT/traffic.json", "w"), indent=1, sort_keys=True)
print(open("$OUT/traffic.json").read())
PY
cp $OUT/traffic.json $R/profiles/traffic.json  # (on the box's copy of the tree: the bench line below carries the traffic of THIS kernel source)
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/stats.log 2>&1 ); echo "stats rc=$?"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
scripts/pmc_cmd.sh ${TAG}_spp64 sq1,sq2,mix,flow,cache,grbm scripts/sched_probe.py spheres > $OUT/pmc_spp64.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_${TAG}_spp64/summary.json $OUT/pmc_spp64.json
for f in $OUT/bench.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['roofline']['frac']); [print(' ', o) for o in d.get('other_configs', [])]")"; done
