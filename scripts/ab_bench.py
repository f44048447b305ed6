"""A/B timing of kernel build variants on the GPU box, interleaved rounds in fresh
processes (cdna guide rule 24). usage: ab_bench.py name=FLAGS ... [--work a,b] [--rounds N]"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import build as B

variants, work, rounds = {}, "spheres64", 2
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--work": work = args.pop(0)
    elif a == "--rounds": rounds = int(args.pop(0))
    else:
        name, _, flags = a.partition("=")
        variants[name] = flags.split() if flags else []
# "name=@path/to/libsrt_hip.so" uses a prebuilt library (e.g. an older revision built before gpurun)
libs = {n: (f[0][1:] if f and f[0].startswith("@") else str(B.build_variant(n, f))) for n, f in variants.items()}
res = {n: [] for n in variants}
for r in range(rounds):
    for n, lib in libs.items():
        env = dict(os.environ, SRT_LIB=lib)
        out = subprocess.run([sys.executable, str(ROOT / "scripts/time_one.py"), work, "3"], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(n, "FAILED", out.stderr[-500:]); continue
        res[n].append(json.loads(out.stdout.strip().splitlines()[-1]))
for n, rs in res.items():
    for w in work.split(","):
        ms = [x[w]["ms"] for x in rs if w in x]
        cs = {x[w]["checksum"] for x in rs if w in x}
        print(f"{n:24s} {w:10s} min {min(ms):9.3f} ms  all {ms}  checksum {cs}")
