"""Builds -D variants of the library in parallel on the CPU box (hipcc cross-compiles; the .so files travel with gpurun):
    python scripts/build_variants.py "name=-DFLAG ..." ...     -> simple-raytracer_amd/lib/variants/<name>/libsrt_hip.so
e.g. the probes of scripts/r04_issue_cost.sh:  for k in 1..22:  "k$k=-DSRT_DIAG -DSRT_DUMMY_KIND=$k",  "d_base=-DSRT_DIAG";
the builds the probe scripts expect:  "diag=-DSRT_DIAG -DSRT_DEV_KNOBS"  "regions=-DSRT_REGION_COUNT"."""
import sys, time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import build as B

variants = {}
for a in sys.argv[1:]:
    name, _, flags = a.partition("=")
    variants[name] = flags.split()


def make(item):
    t0 = time.time()
    B.build_variant(*item)
    return item[0], time.time() - t0


with ThreadPoolExecutor(4) as ex:
    for name, dt in ex.map(make, variants.items()):
        print(name, f"{dt:.0f}s", flush=True)
