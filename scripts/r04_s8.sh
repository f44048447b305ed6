cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s8.log
: > $OUT
for w in 21 20 19; do
echo "== waves $w" >> $OUT
SRT_WAVES_PER_CU=$w timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1 | cut -c1-120 >> $OUT
done
timeout -k 10 300 bash scripts/pmc_cmd.sh r04_masks sq1,sq2 scripts/sched_probe.py spheres >> $OUT 2>&1
cat $OUT
