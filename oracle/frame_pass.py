"""oracle/frame_pass.py -- TEST INFRASTRUCTURE (checker only).

Drives a CPU implementation of the single-frame kernels of Assign01 / Assign04 / Assign07 in the order the
reference host enqueues them (A01 code.js:166-269 compute; A04 code.js:553-577 / A07 code.js:603-628 computeTri:
initTrace, meshTrace, read back).  Implementations: oracle/_ref/libref_a0N.so (the reference's own code.cl compiled
for x86, build container only) and oracle/liboracle.so (our restatement).
"""
import ctypes as C
import os

import numpy as np

import a10_pass as A

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib(kind, assign):
    if kind == "ref":
        lib = C.CDLL(os.path.join(HERE, "_ref", f"libref_a{assign:02d}.so"))
        A.set_hw_tables(lib, "ref_")
        return lib, f"ref_a{assign:02d}_"
    if kind == "count":   # the single-threaded build with the fp32-operation counters (make -C oracle count; count_flops.py)
        lib = C.CDLL(os.path.join(HERE, "liboracle_count.so"))
        A.set_hw_tables(lib, "oracle_")
        return lib, f"oracle_a{assign:02d}_"
    lib = C.CDLL(os.path.join(HERE, "liboracle.so"))
    A.set_hw_tables(lib, "oracle_")
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_set_threads.restype = None
    lib.oracle_set_threads(A.cpu_budget())      # threads = the CPUs this process may really use (cgroup quota), see a10_pass.cpu_budget
    return lib, f"oracle_a{assign:02d}_"


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a.ctypes.data_as(C.POINTER(C.c_float)), a


class Frame:
    """Packed inputs of one frame job (the JSON of gen/ref_host_dump_frame.js or of our JS host's `frame` packer)."""

    def __init__(self, d):
        self.d = d
        self.assign, self.width, self.height = int(d["assign"]), int(d["width"]), int(d["height"])
        self.cam = np.asarray(d["cam"], np.float32)
        self.mol = "atoms" in d                                   # A07 molecule mode (parsePDB + splitMolData + molTrace)
        if self.mol:
            self.bounds = np.asarray(d["bounds"], np.float32)
            self.s_size = int(d["s_size"])
            self.atoms = np.asarray(d["atoms"], np.float32)
            self.mindex, self.mcolor = np.asarray(d["mindex"], np.uint32), np.asarray(d["mcolor"], np.float32)
        elif self.assign != 1:
            self.bounds = np.asarray(d["bounds"], np.float32)
            self.t_size = int(d["t_size"])
            self.pos, self.normal = np.asarray(d["pos"], np.float32), np.asarray(d["normal"], np.float32)
            self.mindex, self.mcolor = np.asarray(d["mindex"], np.uint32), np.asarray(d["mcolor"], np.float32)
        if self.assign == 7:
            self.n_slabs, self.slab_size = int(d["n_slabs"]), np.asarray(d["slab_size"], np.uint32)


def run_frame(kind, fr):
    """Returns (pixels uint8 [H*W,4], rays structured [H*W] or None)."""
    lib, pre = _lib(kind, fr.assign)
    w, h = fr.width, fr.height
    gx, gy = A._ceil(w, 8), A._ceil(h, 8)          # getLocalWS(2, 64) = [8, 8]
    pixels = np.zeros((w * h, 4), np.uint8)
    cp, _c = _f(fr.cam)
    sz, vp, fp, u = C.c_size_t, C.c_void_p, C.POINTER(C.c_float), C.c_uint
    if fr.assign == 1:
        f = getattr(lib, pre + "raytrace"); f.argtypes = [vp, fp, sz, sz]; f.restype = None
        f(_p(pixels), cp, w, h)                    # exactly cols x rows: the reference kernel has no range check
        return pixels, None
    rays = np.zeros(w * h, A.RAY_DT)
    if fr.assign == 4:
        f = getattr(lib, pre + "initTrace"); f.argtypes = [vp, fp, vp, sz, sz]; f.restype = None
        f(_p(pixels), cp, _p(rays), gx, gy)
        f = getattr(lib, pre + "meshTrace"); f.argtypes = [vp, fp, vp, u, vp, vp, vp, vp, sz, sz]; f.restype = None
        f(_p(pixels), cp, _p(rays), fr.t_size, _p(fr.pos), _p(fr.normal), _p(fr.mindex), _p(fr.mcolor), gx, gy)
    elif fr.mol:
        bp, _b = _f(fr.bounds)
        f = getattr(lib, pre + "initTrace"); f.argtypes = [vp, fp, vp, fp, sz, sz]; f.restype = None
        f(_p(pixels), cp, _p(rays), bp, gx, gy)
        f = getattr(lib, pre + "molTrace"); f.argtypes = [vp, fp, vp, u, vp, vp, vp, fp, u, vp, sz, sz]; f.restype = None
        f(_p(pixels), cp, _p(rays), fr.s_size, _p(fr.atoms), _p(fr.mindex), _p(fr.mcolor), bp, fr.n_slabs, _p(fr.slab_size), gx, gy)
    else:
        bp, _b = _f(fr.bounds)
        f = getattr(lib, pre + "initTrace"); f.argtypes = [vp, fp, vp, fp, sz, sz]; f.restype = None
        f(_p(pixels), cp, _p(rays), bp, gx, gy)
        f = getattr(lib, pre + "meshTrace"); f.argtypes = [vp, fp, vp, u, vp, vp, vp, vp, fp, u, vp, sz, sz]; f.restype = None
        f(_p(pixels), cp, _p(rays), fr.t_size, _p(fr.pos), _p(fr.normal), _p(fr.mindex), _p(fr.mcolor), bp, fr.n_slabs, _p(fr.slab_size), gx, gy)
    return pixels, rays


# ---- the same jobs on the reference binaries for the MI355X (oracle/_ref/a0N_gfx950.hsaco, oracle/ref_gpu.py) ------------------------
FRAME_KERNEL_ARGS = {   # argument kinds in the kernels' own order: B buffer, U uint, V float16, X AABB
    1: {"raytrace": "BV"},                                                          # A01 code.cl:116
    4: {"initTrace": "BVB", "meshTrace": "BVBUBBBB"},                               # A04 code.cl:204, 262
    7: {"initTrace": "BVBX", "meshTrace": "BVBUBBBBXUB", "molTrace": "BVBUBBBXUB"},  # A07 code.cl:311, 475, 337
}


def run_frame_gpu(fr, device=0, default_build=False):
    """run_frame on the reference's own kernels as AMD's OpenCL toolchain builds them, on the device.  Same launch shapes as the reference
    host: 2-D NDRange, local [8, 8], globals padded (A01: exactly cols x rows, its kernel has no range check).  default_build: the code object built
    WITHOUT -cl-fp32-correctly-rounded-divide-sqrt (program.build() without options, as the reference's own host builds it: the second contract)."""
    import ref_gpu as G
    mod = G.GpuModule(os.path.join(HERE, "_ref", f"a{fr.assign:02d}_gfx950{'_default' if default_build else ''}.hsaco"), FRAME_KERNEL_ARGS[fr.assign], device)
    w, h = fr.width, fr.height
    g = [A._ceil(w, 8), A._ceil(h, 8)]
    pixels = np.zeros((w * h, 4), np.uint8)
    cam = np.ascontiguousarray(fr.cam, np.float32).tobytes()
    B = mod.buf
    try:
        if fr.assign == 1:
            assert w % 8 == 0 and h % 8 == 0
            mod.launch("raytrace", [B(pixels), cam], [w, h], [8, 8])
            mod.flush()
            return pixels, None
        rays = np.zeros(w * h, A.RAY_DT)
        if fr.assign == 4:
            mod.launch("initTrace", [B(pixels), cam, B(rays)], g, [8, 8])
            mod.launch("meshTrace", [B(pixels), cam, B(rays), fr.t_size, B(fr.pos), B(fr.normal), B(fr.mindex), B(fr.mcolor)], g, [8, 8])
        else:
            bound = np.ascontiguousarray(fr.bounds, np.float32).tobytes()
            mod.launch("initTrace", [B(pixels), cam, B(rays), bound], g, [8, 8])
            if fr.mol:
                mod.launch("molTrace", [B(pixels), cam, B(rays), fr.s_size, B(fr.atoms), B(fr.mindex), B(fr.mcolor), bound, fr.n_slabs, B(fr.slab_size)], g, [8, 8])
            else:
                mod.launch("meshTrace", [B(pixels), cam, B(rays), fr.t_size, B(fr.pos), B(fr.normal), B(fr.mindex), B(fr.mcolor), bound, fr.n_slabs, B(fr.slab_size)], g, [8, 8])
        mod.flush()
        return pixels, rays
    finally:
        mod.release()
