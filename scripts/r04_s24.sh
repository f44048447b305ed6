cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s24.log
: > $OUT
for v in nl1 nl0 nl1 nl0; do
echo "== $v" >> $OUT
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 200 python scripts/time_full.py spheres 2 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(v['kernel_ms_min'],v['checksum']) for k,v in d.items()})" >> $OUT
done
cat $OUT
