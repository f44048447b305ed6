#!/bin/bash
# usage: scripts/pmc_cmd.sh <tag> <group,group,...> <python script + args...>   (run on the GPU box via gpurun)
# One rocprofv3 --pmc pass per counter group (never combined with tracing other than --kernel-trace); prints and
# saves per-launch averages of every counter for the srt_* kernels into gpurun_out/pmc_<tag>/summary.json.
set -u
TAG=$1; GROUPS_=$2; shift 2
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A G
G[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
G[sq2]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
G[mix]="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64"
G[flow]="SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
G[cache]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQ_LDS_BANK_CONFLICT"
G[lds]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
G[vmem]="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"
G[ta1]="TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"
G[ta2]="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
G[ta3]="TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
G[td]="TD_TD_BUSY_sum TD_TC_STALL_sum"
G[tcp2]="TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
G[tcp3]="TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum"
G[tcp4]="TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_WRITE_REQ_sum"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
G[grbm]="GRBM_GUI_ACTIVE GRBM_COUNT"
for name in ${GROUPS_//,/ }; do
  rocprofv3 --kernel-trace --pmc ${G[$name]} --output-format csv -d $OUT/$name -- python3 "$R/$1" "${@:2}" > $OUT/$name.log 2>&1
  echo "$name rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"].split("(")[0]
        if "srt_" not in kn: continue
        a = agg[kn][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for kn, d in agg.items():
        for k, (s, n) in d.items():
            res[kn][k] = s / n
            res[kn]["launches"] = n
json.dump(res, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
for kn, d in res.items():
    if "trace" in kn: print(kn, json.dumps(d, sort_keys=True))
PY
