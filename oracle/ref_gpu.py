"""oracle/ref_gpu.py -- TEST INFRASTRUCTURE (checker only; runs on the GPU box).

The reference's own OpenCL C kernels, compiled for gfx950 by AMD's OpenCL C toolchain -- the ROCm clang in OpenCL mode,
linked against AMD's own OpenCL built-in library (/opt/rocm/amdgcn/bitcode/opencl.bc, ocml.bc, ockl.bc: the library the
ROCm OpenCL runtime links) -- straight from /root/reference by oracle/Makefile (`make ref_gpu`), into
oracle/_ref/a10_gfx950.hsaco (git-ignored; travels to the GPU box like every other built object; the source does not).
No stand-ins: get_global_id, dot, cross, normalize, length, distance, sqrt, sin, cos, fmin, fmax, min, max, mad, clamp
are AMD's.  The one build option added is -cl-fp32-correctly-rounded-divide-sqrt (OpenCL 1.2 section 5.6.4.2), which
makes `/` and sqrt IEEE-exact and therefore reproducible by a CPU checker; without it AMD's division is a 2.5-ulp
v_rcp_f32 sequence whose bits only the GPU can produce (oracle/_ref/a10_gfx950_default.hsaco is that build, used to
measure how far the two conformant builds diverge).

This module loads such a code object with the HIP module API (ctypes on libamdhip64) and exposes the kernels behind the
same interface as a10_pass.CpuKernels, so the same pass driver / call-trace player runs them ON THE MI355X.
It is the strongest oracle the repo has: `the reference OpenCL output` on this hardware.

rays_per_pixel == 1: initTrace's getRand indexes seeds[get_global_id(0)] inside a 2-D NDRange (A10 code.cl:429 vs
469-470): all rows of a column race on one seed.  On a GPU that order is undefined; initTrace at rpp == 1 is the one
thing this oracle cannot pin (hazard 1).
"""
import ctypes as C
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# argument kinds in the kernels' own order (A10 code.cl:440-1386): B buffer, U uint, F float, V float16, X AABB
KERNEL_ARGS = {
    "sizeofRay": "B", "sizeofPoi": "B", "initAcu": "BU", "initTrace": "BBBXVFFU",
    "sphereTrace": "UBBBBBXU", "triangleTrace": "UBBBBBBXU", "meshTrace": "UBBBBBUXU",
    "lightRender": "BBBVU", "initShadowTrace": "BBUVB", "sphereShadowTrace": "UBBBXU",
    "triangleShadowTrace": "UBBBXU", "sceneRender": "BBBBVU", "bouncePaths": "BBBU", "copyToPixel": "BBFUU",
}
_SIZE_ALIGN = {"B": (8, 8), "U": (4, 4), "F": (4, 4), "V": (64, 64), "X": (32, 16)}   # OpenCL C: float16 aligns to 64, struct of two float3 to 16


def kernarg_layout(kinds):
    """[(offset, size)] of the explicit arguments and the size of the explicit block (natural OpenCL C alignment; checked against
    the code object's own metadata by tests/test_ref_gpu.py)."""
    off, out = 0, []
    for k in kinds:
        size, align = _SIZE_ALIGN[k]
        off = (off + align - 1) // align * align
        out.append((off, size))
        off += size
    return out, off


_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipGetErrorString.restype = C.c_char_p
        _hip.hipMalloc.argtypes = [C.c_void_p, C.c_size_t]                      # sizes beyond 2^31: never leave them to ctypes' int default
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipFree.argtypes = [C.c_void_p]
    return _hip


def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {hip().hipGetErrorString(rc).decode()}")


class DevBuf:
    """device-only memory for runs too large to mirror on the host (the full headline frame: ~95 GB of rays, vertices, shadow rays)"""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        chk(hip().hipMalloc(C.byref(p), max(self.nbytes, 16)), f"hipMalloc({self.nbytes})")
        self.ptr = p.value

    def upload(self, a):
        assert a.nbytes <= self.nbytes and a.flags["C_CONTIGUOUS"]
        chk(hip().hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")

    def download(self, dtype, count):
        out = np.empty(count, dtype)
        assert out.nbytes <= self.nbytes
        chk(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(C.c_void_p(self.ptr))
            self.ptr = 0


class GpuModule:
    """A code object compiled by `make -C oracle ref_gpu`, loaded on the device; `kernel_args` maps kernel name -> argument kinds."""

    on_gpu = True

    def __init__(self, hsaco, kernel_args, device=0):
        self.path = hsaco
        self.kernel_args = kernel_args
        h = hip()
        chk(h.hipSetDevice(device), "hipSetDevice")
        self.mod = C.c_void_p()
        chk(h.hipModuleLoad(C.byref(self.mod), self.path.encode()), f"hipModuleLoad({self.path})")
        self.fn = {}
        self.mirrors = {}   # id(array) -> (array, device pointer)
        for name in kernel_args:
            f = C.c_void_p()
            chk(h.hipModuleGetFunction(C.byref(f), self.mod, name.encode()), f"hipModuleGetFunction({name})")
            self.fn[name] = f

    # ---- buffers
    def buf(self, a):
        """device mirror of a numpy array (uploaded when first seen); returns the device pointer as an int"""
        if isinstance(a, DevBuf):
            return a.ptr
        m = self.mirrors.get(id(a))
        if m is None:
            assert a.flags["C_CONTIGUOUS"]
            p = C.c_void_p()
            chk(hip().hipMalloc(C.byref(p), max(a.nbytes, 16)), "hipMalloc")
            if a.nbytes:
                chk(hip().hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")
            m = self.mirrors[id(a)] = (a, p.value)
        return m[1]

    def upload(self, a):
        """the host array changed (a host write of the call-trace player): refresh its device mirror, if it has one"""
        m = self.mirrors.get(id(a))
        if m is not None and a.nbytes:
            chk(hip().hipMemcpy(C.c_void_p(m[1]), a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")

    def flush(self):
        """copy every mirror back into its numpy array"""
        chk(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
        for a, p in self.mirrors.values():
            if a.nbytes and a.flags["WRITEABLE"]:
                chk(hip().hipMemcpy(a.ctypes.data_as(C.c_void_p), C.c_void_p(p), a.nbytes, 2), "hipMemcpy D2H")

    def release(self):
        for _, p in self.mirrors.values():
            hip().hipFree(C.c_void_p(p))
        self.mirrors = {}

    # ---- launches
    def launch(self, name, values, global_ws, local_ws):
        """values: per argument an int (device pointer / uint), float, or bytes (float16 / AABB by value)"""
        kinds = self.kernel_args[name]
        layout, size = kernarg_layout(kinds)
        raw = bytearray(size)
        for k, (off, sz), v in zip(kinds, layout, values):
            if k == "B":
                struct.pack_into("<Q", raw, off, int(v))
            elif k == "U":
                struct.pack_into("<I", raw, off, int(v) & 0xFFFFFFFF)
            elif k == "F":
                struct.pack_into("<f", raw, off, float(v))
            else:
                b = bytes(v)
                assert len(b) == sz, (name, k, len(b))
                raw[off:off + sz] = b
        kbuf = (C.c_char * size).from_buffer(raw)
        ksz = C.c_size_t(size)
        # HIP_LAUNCH_PARAM_BUFFER_POINTER = 1, HIP_LAUNCH_PARAM_BUFFER_SIZE = 2, HIP_LAUNCH_PARAM_END = 3
        extra = (C.c_void_p * 5)(1, C.cast(kbuf, C.c_void_p), 2, C.cast(C.pointer(ksz), C.c_void_p), 3)
        g = list(global_ws) + [1] * (3 - len(global_ws))
        l = list(local_ws) + [1] * (3 - len(local_ws))
        for gi, li in zip(g, l):
            assert gi % li == 0, "the reference host pads every global size to its local size"
        chk(hip().hipModuleLaunchKernel(self.fn[name], g[0] // l[0], g[1] // l[1], g[2] // l[2], l[0], l[1], l[2], 0, None, None, extra),
            f"hipModuleLaunchKernel({name})")



class GpuRefKernels(GpuModule):
    """The reference's Assign10 kernels on the GPU.  Same call surface as a10_pass.CpuKernels (one method per kernel, the kernel's
    own argument order followed by the padded global size[s]); buffers are device mirrors of numpy arrays handed out by buf()."""

    def __init__(self, hsaco=None, device=0):
        super().__init__(hsaco or os.path.join(HERE, "_ref", "a10_gfx950.hsaco"), KERNEL_ARGS, device)
        for name in KERNEL_ARGS:
            setattr(self, name, (lambda n: (lambda *a: self._call(n, a)))(name))
        self.sizeofRay = lambda: self._sizeof("sizeofRay")
        self.sizeofPoi = lambda: self._sizeof("sizeofPoi")

    def _call(self, name, a):
        kinds = KERNEL_ARGS[name]
        nk = len(kinds)
        vals = []
        for k, v in zip(kinds, a[:nk]):
            if k in "VX":   # ctypes float pointer (a10_pass._f) or bytes
                n = 16 if k == "V" else 8
                v = bytes(v) if isinstance(v, (bytes, bytearray)) else C.string_at(v, 4 * n)
            elif k == "B" and not isinstance(v, int):
                v = v.value if hasattr(v, "value") else int(v)
            vals.append(v)
        gws = [int(x) for x in a[nk:]]
        self.launch(name, vals, gws, [8, 8] if name == "initTrace" else [64])

    def _sizeof(self, name):
        out = np.zeros(4, np.uint32)
        self.launch(name, [self.buf(out)], [1], [1])
        self.flush()
        v = int(out[0])
        hip().hipFree(C.c_void_p(self.mirrors.pop(id(out))[1]))
        return v


# oracle/probe/builtins.cl: one kernel per built-in
BUILTIN_ARGS = {"b_" + k: v for k, v in {
    "sqrt": "BBU", "sin": "BBU", "cos": "BBU", "fabs": "BBU", "f2i": "BBU", "f2u": "BBU", "div": "BBBU", "fmin": "BBBU", "fmax": "BBBU",
    "min": "BBBU", "max": "BBBU", "mad": "BBBBU", "clamp": "BBBBU", "muladd": "BBBBU", "dot": "BBBU", "cross": "BBBU", "length": "BBU",
    "distance": "BBBU", "normalize": "BBU"}.items()}
# floats per element of (a, b, c, out); absent: one float per present argument
BUILTIN_SHAPES = {"dot": (3, 3, 0, 1), "cross": (3, 3, 0, 3), "length": (3, 0, 0, 1), "distance": (3, 3, 0, 1), "normalize": (3, 0, 0, 3), "muladd": (1, 1, 1, 4)}


def load_builtins(device=0):
    return GpuModule(os.path.join(HERE, "_ref", "builtins_gfx950.hsaco"), BUILTIN_ARGS, device)


def available(hsaco=None):
    return os.path.exists(hsaco or os.path.join(HERE, "_ref", "a10_gfx950.hsaco"))
