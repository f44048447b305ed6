# Regime probe (round 4): the sphere kernel at full size with 100 extra instructions of ONE kind per loop iteration
# (build the variants first: python scripts/build_variants.py "d_base=-DSRT_DIAG" "k1=-DSRT_DIAG -DSRT_DUMMY_KIND=1" ... "k22=...")
# (-DSRT_DUMMY_KIND=k builds, lib/variants/k<k>) against the plain build: what an instruction of each kind costs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/issue_cost.log
: > $OUT
for v in d_base k1 k2 k3 k4 k5 k6 k7 k8 k9 k10 k11 d_base k12 k13 k14 k15 k16 k17 k18 k19 k20 k21 k22 d_base; do
  echo "== $v" >> $OUT
  SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/$v/libsrt_hip.so timeout -k 10 120 python scripts/time_full.py spheres 2 2>&1 | tail -1 >> $OUT
done
cut -c1-90 $OUT
