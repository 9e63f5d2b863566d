"""ctypes binding of include/mirt.h -- tooling for tests/, bench.py and smoke().

The product host is JavaScript (2015-raytracing_amd/host/, over the N-API addon); this
module exists because the driver's bench/test contract is Python.  It adds nothing to
the C ABI: every method is one mirt_* call, errors become MirtError carrying
mirt_last_error().  There is NO CPU fallback here or below: without libmirt.so or without
a gfx950 device every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIRT_CONTRACT=default: the library built for the reference's own build options (AMD's default 2.5-ulp division and 3-ulp sqrt: csrc/build.sh,
# DESIGN.md section 2) instead of the correctly rounded contract; MIRT_LIB_PATH: an A/B build
LIB_PATH = os.environ.get("MIRT_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libmirt_default.so" if os.environ.get("MIRT_CONTRACT") == "default" else "libmirt.so")

MEM_READ_WRITE, MEM_WRITE_ONLY, MEM_READ_ONLY = 1, 2, 4
MAX_LIGHTS, MAX_MESHES = 8, 16


class MirtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mirt error {code}: {msg}")
        self.code = code


class _Grid(C.Structure):
    _fields_ = [("prims", C.c_void_p), ("normals", C.c_void_p), ("matid", C.c_void_p), ("cell_offsets", C.c_void_p),
                ("bounds", C.c_float * 8), ("n_slabs", C.c_uint32), ("mesh_matid", C.c_uint32)]


class _GridBuildDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_uint32), ("count", C.c_uint32), ("n_slabs", C.c_uint32),
                ("bounds", C.c_double * 6), ("prims_f64", C.c_void_p)]


class _MeshIngestDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_vertices", C.c_uint32), ("n_corners", C.c_uint32), ("first_corner", C.c_uint32),
                ("model", C.c_float * 16), ("normal_mat", C.c_float * 9), ("positions_f64", C.c_void_p), ("normals_f64", C.c_void_p),
                ("indices_u32", C.c_void_p)]


class _Light(C.Structure):
    _fields_ = [("shadow", C.c_float * 16), ("scene", C.c_float * 16), ("light", C.c_float * 16)]


class _PassDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("rays_per_pixel", C.c_uint32),
                ("row0", C.c_uint32), ("nrows", C.c_uint32), ("bounces", C.c_uint32), ("pass_index", C.c_uint32),
                ("cam", C.c_float * 16), ("scene_bounds", C.c_float * 8), ("focal_length", C.c_float), ("lens_rad", C.c_float),
                ("n_lights", C.c_uint32), ("n_meshes", C.c_uint32),
                ("spheres", C.POINTER(_Grid)), ("triangles", C.POINTER(_Grid)), ("meshes", C.POINTER(_Grid)),
                ("lights", C.POINTER(_Light)),
                ("material", C.c_void_p), ("seeds", C.c_void_p), ("acu", C.c_void_p), ("pixel", C.c_void_p), ("radiance", C.c_void_p)]


_lib = None

# name -> (restype, argtypes): every symbol include/mirt.h declares
SYMBOLS = {
    "mirt_device_count": (C.c_int, []),
    "mirt_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "mirt_version": (C.c_char_p, []),
    "mirt_abi_version": (C.c_int, []),
    "mirt_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mirt_ctx_destroy": (C.c_int, [C.c_void_p]),
    "mirt_last_error": (C.c_char_p, [C.c_void_p]),
    "mirt_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mirt_finish": (C.c_int, [C.c_void_p]),
    "mirt_buf_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint, C.POINTER(C.c_void_p)]),
    "mirt_buf_wrap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mirt_buf_release": (C.c_int, [C.c_void_p]),
    "mirt_buf_size": (C.c_size_t, [C.c_void_p]),
    "mirt_buf_device_ptr": (C.c_void_p, [C.c_void_p]),
    "mirt_buf_write": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]),
    "mirt_buf_read": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]),
    "mirt_program_check": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "mirt_program_dialect": (C.c_int, [C.c_char_p]),
    "mirt_kernel_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "mirt_kernel_release": (C.c_int, [C.c_void_p]),
    "mirt_kernel_num_args": (C.c_int, [C.c_void_p]),
    "mirt_kernel_set_arg": (C.c_int, [C.c_void_p, C.c_uint, C.c_size_t, C.c_void_p]),
    "mirt_kernel_set_arg_buf": (C.c_int, [C.c_void_p, C.c_uint, C.c_void_p]),
    "mirt_kernel_preferred_multiple": (C.c_int, [C.c_void_p]),
    "mirt_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "mirt_render_pass": (C.c_int, [C.c_void_p, C.POINTER(_PassDesc)]),
    "mirt_render_first_pass": (C.c_int, [C.c_void_p, C.POINTER(_PassDesc)]),
    "mirt_pass_deferred": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mirt_ctx_set_exact_only": (C.c_int, [C.c_void_p, C.c_int]),
    "mirt_ctx_set_fusion": (C.c_int, [C.c_void_p, C.c_int]),
    "mirt_ctx_fused_passes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mirt_seed_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]),
    "mirt_zero": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mirt_timer_start": (C.c_int, [C.c_void_p]),
    "mirt_timer_stop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mirt_ctx_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mirt_pass_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mirt_grid_build": (C.c_int, [C.c_void_p, C.POINTER(_GridBuildDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    "mirt_grid_gather_triangles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_double), C.c_float, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mirt_grid_gather_spheres": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mirt_grid_gather_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mirt_debug_divcheck": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]),
    "mirt_capture_begin": (C.c_int, [C.c_void_p]),
    "mirt_capture_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mirt_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mirt_graph_release": (C.c_int, [C.c_void_p]),
    "mirt_debug_prepared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mirt_debug_numerics": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mirt_buf_invalidate": (C.c_int, [C.c_void_p]),
    "mirt_mesh_ingest": (C.c_int, [C.c_void_p, C.POINTER(_MeshIngestDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "mirt_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "mirt_group_size": (C.c_int, [C.c_void_p]),
    "mirt_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "mirt_group_destroy": (C.c_int, [C.c_void_p]),
    "mirt_group_finish": (C.c_int, [C.c_void_p]),
    "mirt_tile_rows": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mirt_gather": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "mirt_group_peer_access": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mirt_gather_route": (C.c_int, [C.c_void_p, C.c_int]),
}


def lib():
    """Load libmirt.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MirtError(-7, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("MIRT_LIB_PATH") and not hasattr(l, name):
                continue   # an A/B build of an older tree (profiles/ab.sh): it lacks the newer entry points, which its runs do not call
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def device_count():
    return lib().mirt_device_count()


class Buffer:
    def __init__(self, ctx, handle, nbytes):
        self.ctx, self.h, self.nbytes = ctx, handle, nbytes

    def write(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        self.ctx._chk(lib().mirt_buf_write(self.h, offset, a.nbytes, a.ctypes.data_as(C.c_void_p), 0))
        return self

    def read(self, dtype, count=None, offset=0):
        dt = np.dtype(dtype)
        n = (self.nbytes - offset) // dt.itemsize if count is None else count
        out = np.empty(n, dt)
        self.ctx._chk(lib().mirt_buf_read(self.h, offset, out.nbytes, out.ctypes.data_as(C.c_void_p), 1))
        return out

    @property
    def device_ptr(self):
        return lib().mirt_buf_device_ptr(self.h)

    def release(self):
        if self.h:
            self.ctx._chk(lib().mirt_buf_release(self.h))
            self.h = None


class Kernel:
    """program.createKernel(name): setArg takes a Buffer or a numpy array, as WebCL takes a typed array."""

    def __init__(self, ctx, name):
        self.ctx, self.name = ctx, name
        h = C.c_void_p()
        ctx._chk(lib().mirt_kernel_get(ctx.h, name.encode(), C.byref(h)))
        self.h = h

    def set_arg(self, i, v):
        if isinstance(v, Buffer):
            self.ctx._chk(lib().mirt_kernel_set_arg_buf(self.h, i, v.h))
        else:
            a = np.ascontiguousarray(v)
            self.ctx._chk(lib().mirt_kernel_set_arg(self.h, i, a.nbytes, a.ctypes.data_as(C.c_void_p)))
        return self

    def set_args(self, *vals):
        for i, v in enumerate(vals):
            if v is not None:
                self.set_arg(i, v)
        return self

    def enqueue(self, global_ws, local_ws=None):
        g = (C.c_size_t * len(global_ws))(*global_ws)
        l = (C.c_size_t * len(local_ws))(*local_ws) if local_ws else None
        self.ctx._chk(lib().mirt_enqueue(self.ctx.h, self.h, len(global_ws), g, l))

    def release(self):
        if self.h:
            self.ctx._chk(lib().mirt_kernel_release(self.h))
            self.h = None


class Context:
    """webcl.createContext(device) + ctx.createCommandQueue()."""

    def __init__(self, device=0, _handle=None):
        if _handle is not None:            # a context owned by a DeviceGroup
            self.h, self.device, self.owned = C.c_void_p(_handle), device, False
            return
        h = C.c_void_p()
        rc = lib().mirt_ctx_create(device, C.byref(h))
        if rc != 0:
            raise MirtError(rc, lib().mirt_last_error(None).decode())
        self.h = h
        self.device = device
        self.owned = True

    def _chk(self, rc):
        if rc != 0:
            raise MirtError(rc, lib().mirt_last_error(self.h).decode())

    def last_error(self):
        return lib().mirt_last_error(self.h).decode()

    def buffer(self, nbytes, flags=MEM_READ_WRITE):
        h = C.c_void_p()
        self._chk(lib().mirt_buf_create(self.h, nbytes, flags, C.byref(h)))
        return Buffer(self, h, nbytes)

    def buffer_from(self, arr, flags=MEM_READ_ONLY):
        a = np.ascontiguousarray(arr)
        return self.buffer(max(a.nbytes, 1), flags).write(a) if a.nbytes else self.buffer(16, flags)

    def wrap(self, device_ptr, nbytes):
        h = C.c_void_p()
        self._chk(lib().mirt_buf_wrap(self.h, C.c_void_p(device_ptr), nbytes, C.byref(h)))
        return Buffer(self, h, nbytes)

    def kernel(self, name):
        return Kernel(self, name)

    def set_stream(self, hip_stream):
        self._chk(lib().mirt_ctx_set_stream(self.h, C.c_void_p(hip_stream)))

    def finish(self):
        self._chk(lib().mirt_finish(self.h))

    def seed_fill(self, buf, first_ray, count, seed_base=0):
        self._chk(lib().mirt_seed_fill(self.h, buf.h, first_ray, count, seed_base))

    def zero(self, buf):
        self._chk(lib().mirt_zero(self.h, buf.h))

    # launch-bound sequences as HIP graphs (mirt.h "launch-bound sequences")
    def capture_begin(self):
        self._chk(lib().mirt_capture_begin(self.h))

    def capture_end(self):
        g = C.c_void_p()
        self._chk(lib().mirt_capture_end(self.h, C.byref(g)))
        return g

    def graph_launch(self, g):
        self._chk(lib().mirt_graph_launch(self.h, g))

    def graph_release(self, g):
        lib().mirt_graph_release(g)

    def timer_start(self):
        self._chk(lib().mirt_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._chk(lib().mirt_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def pass_deferred(self):
        n = C.c_uint64()
        self._chk(lib().mirt_pass_deferred(self.h, C.byref(n)))
        return n.value

    def set_exact_only(self, on=True):
        self._chk(lib().mirt_ctx_set_exact_only(self.h, 1 if on else 0))

    def set_fusion(self, level=2):
        """command-stream fusion (include/mirt.h): 2 = whole executeRender passes issued kernel by kernel run as one fused launch"""
        self._chk(lib().mirt_ctx_set_fusion(self.h, int(level)))

    def fused_passes(self):
        n = C.c_uint64()
        self._chk(lib().mirt_ctx_fused_passes(self.h, C.byref(n)))
        return n.value

    def set_profiling(self, on=True):
        self._chk(lib().mirt_ctx_set_profiling(self.h, 1 if on else 0))

    def pass_timing(self):
        a, b = C.c_float(), C.c_float()
        self._chk(lib().mirt_pass_timing(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def program_check(self, source):
        miss = C.create_string_buffer(1024)
        n = lib().mirt_program_check(self.h, source.encode(), miss, 1024)
        if n < 0:
            self._chk(n)
        return n, miss.value.decode()

    def debug_numerics(self, op, a, b=None):
        """op 0..14: one float per element; 20..27 except 25: float3 arguments as three consecutive floats (include/mirt.h)"""
        a = np.ascontiguousarray(a, np.float32)
        vec = op >= 20 and op != 25
        n = a.size // 3 if vec else a.size
        out_w = 3 if op in (21, 24) else 4 if op == 27 else 1
        da = self.buffer_from(a, MEM_READ_WRITE)
        db = self.buffer_from(np.ascontiguousarray(b, np.float32), MEM_READ_WRITE) if b is not None else None
        do = self.buffer(max(4 * n * out_w, 16))
        self._chk(lib().mirt_debug_numerics(self.h, op, da.h, db.h if db else None, do.h, n))
        out = do.read(np.float32, n * out_w)
        for x in (da, db, do):
            if x:
                x.release()
        return out

    def grid_build(self, kind, prims_f64, bounds6, n_slabs):
        """splitSphereData / splitTriangleData / splitMeshData on the device.  Returns (offsets Buffer, order Buffer, total)."""
        prims = np.ascontiguousarray(prims_f64, np.float64)
        per = 9 if kind else 4
        count = prims.size // per
        pb = self.buffer_from(prims, MEM_READ_WRITE) if count else None
        d = _GridBuildDesc()
        d.struct_size, d.kind, d.count, d.n_slabs = C.sizeof(_GridBuildDesc), kind, count, n_slabs
        d.bounds = (C.c_double * 6)(*[float(x) for x in bounds6])
        d.prims_f64 = pb.h if pb else None
        off, order, total = C.c_void_p(), C.c_void_p(), C.c_uint32()
        try:
            self._chk(lib().mirt_grid_build(self.h, C.byref(d), C.byref(off), C.byref(order), C.byref(total)))
        finally:
            if pb:
                pb.release()
        return Buffer(self, off, (n_slabs ** 3 + 1) * 4), Buffer(self, order, max(total.value * 4, 16)), total.value

    def grid_gather_triangles(self, order, total, pos_f64, nor_f64=None, steps=(), pad_w=0.0):
        pb = self.buffer_from(np.ascontiguousarray(pos_f64, np.float64), MEM_READ_WRITE)
        nb = self.buffer_from(np.ascontiguousarray(nor_f64, np.float64), MEM_READ_WRITE) if nor_f64 is not None else None
        ops = (C.c_int32 * 4)(*([s[0] for s in steps] + [0] * (4 - len(steps))))
        vecs = (C.c_double * 12)(*([float(x) for s in steps for x in s[1]] + [0.0] * (12 - 3 * len(steps))))
        po, no = C.c_void_p(), C.c_void_p()
        try:
            self._chk(lib().mirt_grid_gather_triangles(self.h, order.h, total, pb.h, nb.h if nb else None, len(steps), ops, vecs, pad_w,
                                                       C.byref(po), C.byref(no) if nb else None))
        finally:
            pb.release()
            if nb:
                nb.release()
        return Buffer(self, po, max(total * 48, 16)), (Buffer(self, no, max(total * 48, 16)) if nb else None)

    def grid_gather_spheres(self, order, total, sph_f64):
        sb = self.buffer_from(np.ascontiguousarray(sph_f64, np.float64), MEM_READ_WRITE)
        out = C.c_void_p()
        try:
            self._chk(lib().mirt_grid_gather_spheres(self.h, order.h, total, sb.h, C.byref(out)))
        finally:
            sb.release()
        return Buffer(self, out, max(total * 16, 16))

    def grid_gather_u32(self, order, total, values_u32):
        vb = self.buffer_from(np.ascontiguousarray(values_u32, np.uint32), MEM_READ_WRITE)
        out = C.c_void_p()
        try:
            self._chk(lib().mirt_grid_gather_u32(self.h, order.h, total, vb.h, C.byref(out)))
        finally:
            vb.release()
        return Buffer(self, out, max(total * 4, 16))

    def divcheck(self, mode, seed, count):
        out = self.buffer(16 * 8)
        self._chk(lib().mirt_debug_divcheck(self.h, mode, seed, count, out.h))
        r = out.read(np.uint64, 16)
        out.release()
        return r

    def prepared(self, positions, count):
        """the runtime's prepared copy of a triangle position buffer (records, group spheres, plane list) as bytes"""
        total = C.c_size_t(0)
        self._chk(lib().mirt_debug_prepared(self.h, positions.h, count, None, 0, C.byref(total)))
        out = np.zeros(total.value, np.uint8)
        self._chk(lib().mirt_debug_prepared(self.h, positions.h, count, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(total)))
        return out

    def render_pass(self, desc, fresh=False):
        """fresh: the frame's first pass with initAcu folded in (mirt_render_first_pass): acu is not read."""
        f = lib().mirt_render_first_pass if fresh else lib().mirt_render_pass
        self._chk(f(self.h, C.byref(desc)))

    def destroy(self):
        if self.h:
            lib().mirt_ctx_destroy(self.h)
            self.h = None


def _f(arr, n):
    return (C.c_float * n)(*[float(x) for x in np.asarray(arr, np.float32).ravel()[:n]])


class DeviceScene:
    """Uploads the packed kernel inputs of one scene (the arrays the reference host would
    enqueueWriteBuffer: A10 code.js:1183-1185, 1231-1234, 1276-1278, 1379) and keeps the
    handles.  `s` is a pyhost.scene.PackedScene (or anything with its attributes)."""

    def __init__(self, ctx, s):
        self.ctx, self.s = ctx, s
        self.bufs = []
        up = self._up
        self.sph = self.tri = None
        if s.has_spheres:
            self.sph = dict(prims=up(s.spheres), matid=up(s.s_matid), off=up(s.s_box), bounds=s.sphere_bounds, n=s.n_slabs)
        if s.has_triangles:
            self.tri = dict(prims=up(s.t_pos), normals=up(s.t_normal), matid=up(s.t_matid), off=up(s.t_box),
                            bounds=s.triangle_bounds, n=s.n_slabs)
        self.meshes = [dict(prims=up(m["pos"]), normals=up(m["normal"]), off=up(m["box"]), bounds=m["bounds"],
                            n=m["nslabs"], matid=m["matid"]) for m in s.meshes]
        self.material = up(s.materials)

    def _up(self, arr):
        b = self.ctx.buffer_from(arr)
        self.bufs.append(b)
        return b

    def _grid(self, g, mesh=False):
        out = _Grid()
        out.prims = g["prims"].h
        out.normals = g["normals"].h if "normals" in g else None
        out.matid = None if mesh else g["matid"].h
        out.cell_offsets = g["off"].h
        out.bounds = _f(g["bounds"], 8)
        out.n_slabs = int(g["n"])
        out.mesh_matid = int(g["matid"]) if mesh else 0
        return out

    def pass_desc(self, seeds, acu, pixel=None, radiance=None, pass_index=1, bounces=5, row0=0, nrows=0):
        s = self.s
        d = _PassDesc()
        d.struct_size = C.sizeof(_PassDesc)
        d.width, d.height, d.rays_per_pixel = s.width, s.height, s.rpp
        d.row0, d.nrows, d.bounces, d.pass_index = row0, nrows, bounces, pass_index
        d.cam = _f(s.cam, 16)
        d.scene_bounds = _f(s.bounds, 8)
        d.focal_length, d.lens_rad = s.focal_length, s.lens_rad
        keep = []
        if self.sph:
            g = self._grid(self.sph); keep.append(g); d.spheres = C.pointer(g)
        if self.tri:
            g = self._grid(self.tri); keep.append(g); d.triangles = C.pointer(g)
        if self.meshes:
            arr = (_Grid * len(self.meshes))(*[self._grid(m, mesh=True) for m in self.meshes])
            keep.append(arr); d.meshes = C.cast(arr, C.POINTER(_Grid))
        d.n_meshes = len(self.meshes)
        larr = (_Light * max(1, len(s.lights)))()
        for i, l in enumerate(s.lights):
            larr[i].shadow, larr[i].scene, larr[i].light = _f(l["shadow"], 16), _f(l["scene"], 16), _f(l["light"], 16)
        keep.append(larr)
        d.lights = C.cast(larr, C.POINTER(_Light))
        d.n_lights = len(s.lights)
        d.material = self.material.h
        d.seeds, d.acu = seeds.h, (acu.h if acu else None)   # acu None: a frame's first pass that resolves its pixels itself (include/mirt.h)
        d.pixel = pixel.h if pixel else None
        d.radiance = radiance.h if radiance else None
        d._keep = keep
        return d

    def release(self):
        for b in self.bufs:
            b.release()
        self.bufs = []


class DeviceGroup:
    """mirt_group: N contexts in one process (one per device) + mirt_gather, the one exchange of the path."""

    def __init__(self, devices):
        ids = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = lib().mirt_group_create(ids, len(devices), C.byref(h))
        if rc != 0:
            raise MirtError(rc, lib().mirt_last_error(None).decode())
        self.h = h
        self.contexts = [Context(d, _handle=lib().mirt_group_ctx(h, i)) for i, d in enumerate(devices)]

    def tile_rows(self, height, index):
        r0, n = C.c_uint32(), C.c_uint32()
        lib().mirt_tile_rows(height, len(self.contexts), index, C.byref(r0), C.byref(n))
        return r0.value, n.value

    GATHER_AUTO, GATHER_RCCL, GATHER_COPY = 0, 1, 2

    def gather(self, tiles, tile_bytes, out, root=0, use_rccl=False, transport=None):
        """transport: GATHER_AUTO (RCCL for N > 1 distinct devices, copies otherwise), GATHER_RCCL, GATHER_COPY; use_rccl=True == GATHER_RCCL"""
        n = len(tiles)
        hs = (C.c_void_p * n)(*[t.h for t in tiles])
        bs = (C.c_size_t * n)(*tile_bytes)
        if transport is None:
            transport = self.GATHER_RCCL if use_rccl else self.GATHER_AUTO
        rc = lib().mirt_gather(self.h, hs, bs, n, out.h, root, transport)
        if rc != 0:
            raise MirtError(rc, lib().mirt_last_error(self.contexts[0].h).decode())

    ROUTES = {0: "none", 1: "rccl", 2: "peer", 3: "staged", 4: "local"}

    def routes(self):
        """how the last gather moved each tile (mirt_gather_route)"""
        return [self.ROUTES.get(lib().mirt_gather_route(self.h, i), "?") for i in range(len(self.contexts))]

    def peer_access(self, i, j):
        return lib().mirt_group_peer_access(self.h, i, j)

    def finish(self):
        rc = lib().mirt_group_finish(self.h)
        if rc != 0:
            raise MirtError(rc, lib().mirt_last_error(self.contexts[0].h).decode())

    def destroy(self):
        if self.h:
            lib().mirt_group_destroy(self.h)
            self.h = None
