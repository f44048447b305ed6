#!/bin/bash
# usage: scripts/pmc_run.sh <tag> <bench args...>   (run on the GPU box via gpurun)
# One rocprofv3 invocation per counter group (SQ has 8 slots, TCC 4: MI355X_MICROARCH.md).
set -u
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py "${BARGS[@]}" > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
BARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "srt_trace" not in row["Kernel_Name"]: continue
            k = row["Counter_Name"]; agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
        for k, (s, n) in sorted(agg.items()):
            print(f"{k} per_launch={s/n:.6g} launches={n}")
PY
