cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s20.log
: > $OUT
for v in w2_r1 w2_r16 w2_r32; do
echo "== $v" >> $OUT
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 200 python scripts/time_full.py meshes:bvh,mesh100k:bvh 2 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(v['kernel_ms_min'],v['checksum'],v['iterations']) for k,v in d.items()})" >> $OUT
done
for v in w2_r1_reg w2_r16_reg; do
echo "== $v" >> $OUT
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 200 python scripts/region_probe.py mesh100k_bvh 64 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['regions']
print({k:(v[0], round(v[1]/max(v[0],1),1)) for k,v in r.items() if v[0] and k in ('EXTEND_BVH_STEP','EXTEND_MODEL_1','EXTEND_SUSPEND','REFILL_SCANQ','REFILL_UNPARK','LOOP_HEAD','SHADE_BOUNCE','EXTEND_TRI_EXACT')}, d['kernel_ms_instrumented'])" >> $OUT
done
cat $OUT
