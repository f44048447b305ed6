cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for i in 1 2 3; do timeout -k 10 200 python scripts/bench_stall_probe.py --torch 2>> gpurun_out/r04/s26.err | tail -1; done | tee gpurun_out/r04/bench_stall_probe_torch.jsonl
tail -3 gpurun_out/r04/s26.err
