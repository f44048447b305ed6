cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s14.log
: > $OUT
timeout -k 10 200 python scripts/r04_overlap_probe.py alone 2>&1 | tail -1 >> $OUT
timeout -k 10 200 python scripts/r04_overlap_probe.py behind 2>&1 | tail -1 >> $OUT
cat $OUT
