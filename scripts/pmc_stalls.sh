#!/bin/bash
# usage: scripts/pmc_stalls.sh <tag> <bench args...>   (run on the GPU box via gpurun)
# Instruction-mix and stall-side SQ / SQC counters of srt_trace_kernel, one rocprofv3
# invocation per counter group (SQ has 8 slots: MI355X_MICROARCH.md).
set -u
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS=("$@")
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py "${BARGS[@]}" > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run mix SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64
run flow SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM
run cache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQ_LDS_BANK_CONFLICT
run busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
run misc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM
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
