"""MI355X-native drop-in for the path-tracing hot path of davawen/Simple-Raytracer.

Directory name has a hyphen, so import it through `srt_pkg.load()` at the repo root
(registers this package as `simple_raytracer_amd`).

  records  numpy views of the ABI records (include/srt_types.h)
  scenes   builders for the BASELINE.json configs
  tracer   ctypes binding of the C ABI (include/srt_abi.h) + a Tracer class mirroring
           the reference's include/tracer.hpp; needs the HIP library and a GPU
  build    hipcc recipes for csrc/
"""
from . import records, scenes  # noqa: F401

__all__ = ["records", "scenes"]
