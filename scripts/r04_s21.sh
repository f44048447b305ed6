cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s21.log
: > $OUT
for v in diag hb40 hb48 hb64; do
echo "== $v" >> $OUT
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 200 python scripts/time_full.py meshes,meshes:bvh,mesh100k:bvh 2 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:v['kernel_ms_min'] for k,v in d.items()})" >> $OUT
done
cat $OUT
