cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s19.log
: > $OUT
export SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/w2/libsrt_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_bvh.py -x -q 2>&1 | tail -3 >> $OUT
timeout -k 10 200 python scripts/time_full.py meshes:bvh,mesh100k:bvh 2 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(v['kernel_ms_min'],v['checksum'],v['iterations']) for k,v in d.items()})" >> $OUT
cat $OUT
