"""oracle/a10_pass.py -- TEST INFRASTRUCTURE (checker only).

Drives a CPU implementation of the fourteen Assign10 kernels through one
progressive pass in exactly the order the reference host enqueues them
(A10 code.js:1806-1854 executeRender; preRender :1784-1804 for the one-off
initAcu).  Two implementations expose the same C signatures and can be plugged
in:

  * oracle/_ref/libref_a10.so  prefix "ref_a10_"    the reference's own OpenCL C,
    compiled for x86 (build container only, see oracle/Makefile `ref`)
  * oracle/liboracle.so        prefix "oracle_a10_" our plain-C restatement

Buffers are numpy arrays laid out as the compiled reference lays them out:
Ray 48 B {o@0 d@16 mint@32 maxt@36}, Poi 64 B {p@0 normal@16 atte@32 matId@48},
acu float4, pixel uchar4.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

RAY_DT = np.dtype({"names": ["o", "d", "mint", "maxt"],
                   "formats": [(np.float32, 3), (np.float32, 3), np.float32, np.float32],
                   "offsets": [0, 16, 32, 36], "itemsize": 48})
POI_DT = np.dtype({"names": ["p", "normal", "atte", "matId"],
                   "formats": [(np.float32, 3), (np.float32, 3), (np.float32, 3), np.int32],
                   "offsets": [0, 16, 32, 48], "itemsize": 64})

WAVE = 64  # KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE the product reports (one wavefront)


def make_seeds(total_rays, seed_base=0, first=0):
    """s[id] = 1 + (splitmix32(id ^ 0x9E3779B9 ^ seed_base) mod 2147483646)   (SURVEY 8d config 4).

    The reference seeds with Math.random() (code.js:1140-1146): any int32 in
    [1, 2^31-1] is a legal seed; this closed form makes runs reproducible and
    independent of how rays are sharded over GPUs (ids are global).
    """
    ids = (np.arange(first, first + total_rays, dtype=np.uint64) & 0xFFFFFFFF).astype(np.uint32)
    x = ids ^ np.uint32(0x9E3779B9) ^ np.uint32(seed_base & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (x + np.uint32(0x9E3779B9)).astype(np.uint32)
        z = ((z ^ (z >> np.uint32(16))) * np.uint32(0x85EBCA6B)).astype(np.uint32)
        z = ((z ^ (z >> np.uint32(13))) * np.uint32(0xC2B2AE35)).astype(np.uint32)
        z = (z ^ (z >> np.uint32(16))).astype(np.uint32)
    return (1 + (z % np.uint32(2147483646))).astype(np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.ctypes.data_as(C.POINTER(C.c_float)), a


class CpuKernels:
    """ctypes view of one CPU implementation (ref or oracle)."""

    on_gpu = False
    buf = staticmethod(_p)      # a kernel argument for a numpy buffer: its host pointer

    def flush(self):            # (oracle/ref_gpu.py copies its device mirrors back here)
        pass

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        vp, fp, u, f, sz = C.c_void_p, C.POINTER(C.c_float), C.c_uint, C.c_float, C.c_size_t
        sig = {
            "initAcu": [vp, u, sz],
            "initTrace": [vp, vp, vp, fp, fp, f, f, u, sz, sz],
            "bouncePaths": [vp, vp, vp, u, sz],
            "lightRender": [vp, vp, vp, fp, u, sz],
            "initShadowTrace": [vp, vp, u, fp, vp, sz],
            "sphereTrace": [u, vp, vp, vp, vp, vp, fp, u, sz],
            "triangleTrace": [u, vp, vp, vp, vp, vp, vp, fp, u, sz],
            "meshTrace": [u, vp, vp, vp, vp, vp, u, fp, u, sz],
            "sphereShadowTrace": [u, vp, vp, vp, fp, u, sz],
            "triangleShadowTrace": [u, vp, vp, vp, fp, u, sz],
            "sceneRender": [vp, vp, vp, vp, fp, u, sz],
            "copyToPixel": [vp, vp, f, u, u, sz],
        }
        for name, args in sig.items():
            fn = getattr(self.lib, prefix + name)
            fn.argtypes = args
            fn.restype = None
            setattr(self, name, fn)
        for name in ("sizeofRay", "sizeofPoi"):
            fn = getattr(self.lib, prefix + name)
            fn.argtypes = []
            fn.restype = C.c_uint
            setattr(self, name, fn)


_HW = []


def set_hw_tables(lib, prefix):
    """Hands a CPU checker the measured v_rsq_f32 / v_sqrt_f32 tables of the MI355X (oracle/hw_tables.bin.z, made by
    oracle/probe/hw_probe.hip + make_hw_tables.py): int8[2^24] each.  cl_numerics.h aborts without them."""
    import zlib
    if not _HW:
        raw = zlib.decompress(open(os.path.join(HERE, "hw_tables.bin.z"), "rb").read())
        assert len(raw) == 2 << 24
        _HW.append(C.create_string_buffer(raw, len(raw)))
    fn = getattr(lib, prefix + "set_hw_tables")
    fn.argtypes = [C.c_void_p, C.c_void_p]
    fn.restype = None
    base = C.addressof(_HW[0])
    fn(base, base + (1 << 24))


def load_ref():
    k = CpuKernels(os.path.join(HERE, "_ref", "libref_a10.so"), "ref_a10_")
    set_hw_tables(k.lib, "ref_")
    return k


def cpu_budget():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a 256-thread host can hand a container
    sixteen CPUs' worth of time; 128 OpenMP threads inside that quota only fight each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                        # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(np.ceil(q / p))))
        except (OSError, ValueError):
            pass
    return n


def load_oracle():
    k = CpuKernels(os.path.join(HERE, "liboracle.so"), "oracle_a10_")
    k.lib.oracle_set_threads.argtypes = [C.c_int]
    k.lib.oracle_set_threads.restype = None
    k.lib.oracle_set_threads(cpu_budget())
    set_hw_tables(k.lib, "oracle_")
    return k


def _ceil(n, m):
    return (n + m - 1) // m * m


class Scene:
    """The packed kernel inputs of one scene (the JSON of gen/ref_host_dump.js or of our own JS host)."""

    def __init__(self, d):
        self.d = d
        self.width, self.height, self.rpp = d["width"], d["height"], d["rays_per_pixel"]
        self.n_slabs = d.get("n_slabs", 1)
        f32 = lambda k: np.asarray(d[k], dtype=np.float32)
        u32 = lambda k: np.asarray(d[k], dtype=np.uint32)
        self.cam, self.bounds = f32("cam"), f32("bounds")
        self.focal_length, self.lens_rad = float(d["focal_length"]), float(d["lens_rad"])
        self.has_spheres = d.get("n_spheres", 0) > 0
        self.has_triangles = d.get("n_triangles", 0) > 0
        if self.has_spheres:
            self.spheres, self.s_matid, self.s_box, self.sphere_bounds = f32("spheres"), u32("s_matid"), u32("s_box"), f32("sphere_bounds")
        if self.has_triangles:
            self.t_pos, self.t_normal, self.t_matid, self.t_box, self.triangle_bounds = (
                f32("t_pos"), f32("t_normal"), u32("t_matid"), u32("t_box"), f32("triangle_bounds"))
        self.meshes = [dict(pos=np.asarray(m["pos"], np.float32), normal=np.asarray(m["normal"], np.float32),
                            box=np.asarray(m["box"], np.uint32), matid=int(m["matid"]),
                            bounds=np.asarray(m["bounds"], np.float32), nslabs=int(m["nslabs"]))
                       for m in d.get("meshes", [])]
        self.lights = [dict(shadow=np.asarray(l["shadow"], np.float32), scene=np.asarray(l["scene"], np.float32),
                            light=np.asarray(l["light"], np.float32)) for l in d["lights"]]
        self.materials = f32("materials")

    @property
    def total_rays(self):
        return self.width * self.height * self.rpp


class PassState:
    def __init__(self, scene, seeds):
        n = scene.total_rays
        self.rays = np.zeros(n, RAY_DT)
        self.pois = np.zeros(n, POI_DT)
        self.shadow = np.zeros(n, RAY_DT)
        self.acu = np.zeros((n, 4), np.float32)
        self.pixel = np.zeros((scene.width * scene.height, 4), np.uint8)
        self.seeds = np.array(seeds, dtype=np.int32, copy=True)
        self.passes = 1

    def snapshot(self, names=("rays", "pois", "shadow", "acu", "seeds")):
        return {k: getattr(self, k).copy() for k in names}


def run_pass(k, sc, st, bounces=5, checkpoints=None, init_acu=True, rows=None):
    """One executeRender() (code.js:1806-1854).  `checkpoints`: optional dict that
    receives snapshots after the primary segment ('primary') and each bounce.
    `rows`: launch the 2-D initTrace over the first `rows` image rows only and every 1-D kernel over their rays -- the camera still
    says sc.height rows, so these ARE rays 0 .. width*rows*rpp-1 of the full frame (a top band of a frame too big to hold whole)."""
    h = sc.height if rows is None else rows
    n = sc.width * h * sc.rpp
    g1 = _ceil(n, WAVE)
    B = k.buf          # host pointer (CPU kernels) or device mirror (oracle/ref_gpu.py)
    if init_acu:  # preRender -> prepareInitAcu (code.js:1078-1099), once per render
        k.initAcu(B(st.acu), n, g1)

    bp, _b = _f(sc.bounds)
    cp, _c = _f(sc.cam)
    # getLocalWS(2, ...) with a multiple of 64 -> [8, 8]  (code.js:661-663)
    k.initTrace(B(st.seeds), B(st.rays), B(st.pois), bp, cp, sc.focal_length, sc.lens_rad, sc.rpp,
                _ceil(sc.width, 8), _ceil(h, 8))

    def closest():
        if sc.has_spheres:
            p, _k = _f(sc.sphere_bounds)
            k.sphereTrace(n, B(st.pois), B(st.rays), B(sc.spheres), B(sc.s_matid), B(sc.s_box), p, sc.n_slabs, g1)
        if sc.has_triangles:
            p, _k = _f(sc.triangle_bounds)
            k.triangleTrace(n, B(st.pois), B(st.rays), B(sc.t_pos), B(sc.t_normal), B(sc.t_matid), B(sc.t_box), p, sc.n_slabs, g1)
        for m in sc.meshes:
            p, _k = _f(m["bounds"])
            k.meshTrace(n, B(st.pois), B(st.rays), B(m["pos"]), B(m["normal"]), B(m["box"]), m["matid"], p, m["nslabs"], g1)

    def direct():
        for l in sc.lights:
            p, _k = _f(l["shadow"])
            k.initShadowTrace(B(st.shadow), B(st.pois), n, p, B(st.seeds), g1)
            if sc.has_spheres:
                p, _k = _f(sc.sphere_bounds)
                k.sphereShadowTrace(n, B(st.shadow), B(sc.spheres), B(sc.s_box), p, sc.n_slabs, g1)
            if sc.has_triangles:
                p, _k = _f(sc.triangle_bounds)
                k.triangleShadowTrace(n, B(st.shadow), B(sc.t_pos), B(sc.t_box), p, sc.n_slabs, g1)
            for m in sc.meshes:
                p, _k = _f(m["bounds"])
                k.triangleShadowTrace(n, B(st.shadow), B(m["pos"]), B(m["box"]), p, m["nslabs"], g1)
            p, _k = _f(l["scene"])
            k.sceneRender(B(st.acu), B(st.pois), B(st.shadow), B(sc.materials), p, n, g1)

    closest()
    for l in sc.lights:
        p, _k = _f(l["light"])
        k.lightRender(B(st.pois), B(st.rays), B(st.acu), p, n, g1)
    direct()
    if checkpoints is not None:
        k.flush()
        checkpoints["primary"] = st.snapshot()
    for j in range(bounces):
        k.bouncePaths(B(st.pois), B(st.rays), B(st.seeds), n, g1)
        closest()
        direct()
        if checkpoints is not None and j == 0:
            k.flush()
            checkpoints["bounce1"] = st.snapshot()
    m = np.float32(1.0 / (sc.rpp * st.passes))  # code.js:1412: double division, narrowed by Float32Array
    k.copyToPixel(B(st.pixel), B(st.acu), float(m), sc.width * h, sc.rpp, _ceil(sc.width * h, WAVE))
    st.passes += 1
    k.flush()
    return st


def radiance_sums(acu, rpp):
    """Per-pixel sequential fp32 sum of the per-ray accumulators (copyToPixel's order, code.cl:1377-1380)."""
    a = acu.reshape(-1, rpp, 4)
    s = np.zeros((a.shape[0], 4), np.float32)
    for i in range(rpp):
        s = (s + a[:, i, :]).astype(np.float32)
    return s
