cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/s9.log
: > $OUT
for v in cur d_base; do
  if [ $v = cur ]; then unset SRT_LIB; else export SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so; fi
  timeout -k 10 300 bash scripts/pmc_cmd.sh r04_full_$v grbm,sq1 scripts/time_full.py spheres 2 >> $OUT 2>&1
done
python3 - <<'PY' >> $OUT
import csv,glob
for v in ['cur','d_base']:
    for g in ['grbm','sq1']:
        fs=glob.glob(f'gpurun_out/pmc_r04_full_{v}/{g}/**/*kernel_trace.csv',recursive=True)
        cs=glob.glob(f'gpurun_out/pmc_r04_full_{v}/{g}/**/*counter_collection.csv',recursive=True)
        dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(fs[0])) if 'trace' in r['Kernel_Name']]
        ctr={}
        for r in csv.DictReader(open(cs[0])):
            if 'trace' in r['Kernel_Name']: ctr.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        print(v,g,'dur ms',[round(d,2) for d in dur],{k:[round(x/1e6,1) for x in vv] for k,vv in ctr.items() if k in('GRBM_GUI_ACTIVE','SQ_BUSY_CYCLES','SQ_WAVE_CYCLES')})
PY
cat $OUT
