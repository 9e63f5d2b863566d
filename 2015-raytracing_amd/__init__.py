"""2015-raytracing_amd -- MI355X-native device runtime for the Assign10 path tracer of
eaymerich/2015-RayTracing.

  csrc/    hand-written HIP kernels for gfx950 + the C ABI (include/mirt.h) -> libmirt.so
  host/    the JavaScript host (Node): WebCL-shaped binding over the N-API addon, scene
           loader, grid builders, pass drivers -- the mirror of the reference's code.js
  pyhost/  ctypes binding + pass drivers used by tests/, bench.py and smoke()

The directory name is not a Python identifier; `__graft_entry__.load_package()` registers it
as the module `raytracing_amd`.
"""
__version__ = "0.1"
