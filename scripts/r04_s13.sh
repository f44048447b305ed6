cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s13.log
: > $OUT
echo "== product" >> $OUT
timeout -k 10 200 python scripts/time_full.py mesh100k 1 2>&1 | tail -1 | cut -c1-300 >> $OUT
for v in dev diag; do
  echo "== $v" >> $OUT
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 200 python scripts/time_full.py mesh100k 1 2>&1 | tail -1 | cut -c1-300 >> $OUT
done
cat $OUT
