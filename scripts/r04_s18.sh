cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s18.log
: > $OUT
for v in wr0 wr1 wr0 wr1; do
  echo "== $v" >> $OUT
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 200 python scripts/time_full.py meshes:bvh,mesh100k:bvh 2 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(v['kernel_ms_min'],v['checksum']) for k,v in d.items()})" >> $OUT
done
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/wr1r/libsrt_hip.so timeout -k 10 200 python scripts/region_probe.py mesh100k_bvh 64 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['regions']
print({k:(v[0], round(v[1]/max(v[0],1),1)) for k,v in r.items() if v[0] and k in ('EXTEND_BVH_STEP','EXTEND_BVH_LEAF','EXTEND_MODEL_1','LOOP_HEAD','EXTEND_TRI_EXACT','EXTEND_TRI_DIV')}, d['kernel_ms_instrumented'])" >> $OUT
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/wr1/libsrt_hip.so timeout -k 10 400 python -m pytest tests/test_gpu_bvh.py -x -q 2>&1 | tail -2 >> $OUT
cat $OUT
