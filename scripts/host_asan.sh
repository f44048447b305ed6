#!/bin/bash
# usage: scripts/host_asan.sh   (no GPU) -- the HOST side of the library (BVH builder and its wide folding, scene preparation, the host-only ABI
# entry points) under AddressSanitizer + UndefinedBehaviorSanitizer: builds lib/variants/asan with -Xarch_host -fsanitize=address,undefined
# (device code is compiled as always; GPU sanitizers are not available on this pool) and runs the CPU tests that go through the library.
set -eu
cd "$(dirname "$0")/.."
python3 - <<'PY'
import subprocess, srt_pkg
srt_pkg.load()
from simple_raytracer_amd import build as B
out = B.LIBDIR / "variants" / "asan"
out.mkdir(parents=True, exist_ok=True)
flags = [f for f in B.FLAGS if not f.startswith("-Rpass")]
san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-shared-libsan"]
subprocess.run([B.hipcc(), *flags, *san, "-shared", "-o", str(out / "libsrt_hip.so")] + [str(B.CSRC / s) for s in B.SOURCES] + ["-ldl"], check=True, capture_output=True)
PY
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SRT_LIB=$PWD/simple-raytracer_amd/lib/variants/asan/libsrt_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 \
  python3 -m pytest tests/test_bvh_host.py tests/test_abi_host.py tests/test_host_mirror.py -q -m "not gpu" 2>&1 | tee /tmp/host_asan.log | tail -3
if grep -q "runtime error\|AddressSanitizer" /tmp/host_asan.log; then echo "SANITIZER REPORTS ABOVE"; exit 1; fi
echo "no sanitizer reports"
