"""The pin: the reference's own OpenCL C, compiled by AMD's OpenCL toolchain, run on the MI355X.

oracle/_ref/a10_gfx950.hsaco is the reference's Assign10 code.cl compiled for gfx950 by the ROCm clang in OpenCL mode and
linked against AMD's OpenCL C built-in library (oracle/Makefile `ref_gpu`; the text untouched, one option:
-cl-fp32-correctly-rounded-divide-sqrt).  oracle/_ref/builtins_gfx950.hsaco holds one probe kernel per built-in, same
toolchain, same options (oracle/probe/builtins.cl).  oracle/ref_gpu.py launches both through the HIP module API.

GPU tests (the code objects travel, the reference's source does not):
  * the CPU model of the built-ins (oracle/cl_numerics.h: fused dot / cross, v_rsq_f32 / v_sqrt_f32 from tables measured on the
    device, v_min / v_max / v_med3, ocml's sin / cos) == AMD's library, argument by argument, bit for bit;
  * the HIP kernels' numerics layer (csrc/pt_numerics.hpp) == AMD's library likewise;
  * the reference kernels on the device == the committed fixtures (made in the container by the x86 twin of the same build,
    oracle/_ref/libref_a10.so) bit for bit, for every fixture with rays_per_pixel >= 4 (at one ray per pixel initTrace races on
    seeds[col]) -- which is what makes every other parity test in this suite a comparison with the reference
    binary, not with a restatement.
CPU tests: the launcher's kernel-argument layout equals the code object's own metadata (container only: needs llvm-readelf).
"""
import ctypes as C
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import a10_pass as A
import ref_gpu as G
from conftest import FULL_CASES, ROOT, canon, load_fixture

HSACO = os.path.join(ROOT, "oracle", "_ref", "a10_gfx950.hsaco")
BUILTINS = os.path.join(ROOT, "oracle", "_ref", "builtins_gfx950.hsaco")
N = 1 << 20
SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 1.17549435e-38, 3.4e38, -3.4e38, 0.5, 255.0, 256.0, 2.0, 1e-20,
                     1e20, 1e-30, 1e30, 2147483648.0, 4294967296.0, -2147483648.0, 0.99999994, 1.0000001], np.float32)
# HIP-side op numbers of mirt_debug_numerics for the same built-ins (include/mirt.h)
HIP_OPS = {"div": 0, "sqrt": 1, "sin": 2, "cos": 3, "min": 6, "max": 7, "fmin": 8, "fmax": 9, "f2i": 13, "f2u": 14, "dot": 20, "cross": 21,
           "length": 22, "distance": 23, "normalize": 24, "clamp": 25, "mad": 26, "muladd": 27}


def inputs(op, dist, seed):
    rng = np.random.default_rng(seed)
    na, nb, nc, no = G.BUILTIN_SHAPES.get(op, (1, 1 if len(G.BUILTIN_ARGS["b_" + op]) >= 4 else 0, 1 if len(G.BUILTIN_ARGS["b_" + op]) >= 5 else 0, 1))

    def rnd(n):
        if dist == "bits":
            x = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32).copy()
        elif dist == "unit":
            x = rng.uniform(-1, 1, n).astype(np.float32)
        elif dist == "ang":
            x = rng.uniform(-2.5, 2.5, n).astype(np.float32)
        else:
            x = (rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
        x[:len(SPECIALS)] = SPECIALS
        return x
    a = rnd(N * na)
    b = np.roll(rnd(N * nb), 7) if nb else None
    c = np.roll(rnd(N * nc), 13) if nc else None
    if op in ("sin", "cos"):
        a[np.isfinite(a) & (np.abs(a) >= 131072.0)] = 1.0          # the CPU model does not restate ocml's Payne-Hanek path
    if op == "clamp":                                                # clamp(x, 0, hi), hi in {1, 255}: the two uses (A10 code.cl:1352-1353, 1383)
        u = a.view(np.uint32)
        u[(u & 0x7FFFFFFF) > 0x7F800000] |= 0x00400000              # quiet NaNs only: arithmetic never makes a signalling one
        b = np.zeros(N, np.float32)
        c = np.where(rng.random(N) < 0.5, np.float32(1.0), np.float32(255.0)).astype(np.float32)
    return a, b, c, (na, nb, nc, no)


def amd(mod, op, a, b, c, no):
    out = np.zeros(N * no, np.float32)
    args = [mod.buf(a)] + ([mod.buf(b)] if b is not None else []) + ([mod.buf(c)] if c is not None else []) + [mod.buf(out), N]
    mod.launch("b_" + op, args, [N], [64])
    mod.flush()
    mod.release()
    return out


def first_diff(tag, got, want, a):
    bad = np.flatnonzero(canon(got) != canon(want))
    if bad.size:
        i = int(bad[0])
        raise AssertionError(f"{tag}: {bad.size} of {got.size} results differ; first at {i}: {got.view(np.uint32)[i]:#010x} != {want.view(np.uint32)[i]:#010x}")


needs_gpu_ref = pytest.mark.skipif(not (os.path.exists(HSACO) and os.path.exists(BUILTINS)), reason="oracle/_ref/*.hsaco not built (make -C oracle ref_gpu, build container)")


@pytest.fixture(scope="module")
def builtins_mod():
    return G.load_builtins()


@pytest.mark.gpu
@needs_gpu_ref
@pytest.mark.parametrize("op", sorted(k[2:] for k in G.BUILTIN_ARGS))
def test_cpu_model_equals_amd_opencl_library(builtins_mod, op):
    orc = A.load_oracle()
    orc.lib.oracle_bi_eval.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    orc.lib.oracle_bi_eval.restype = None
    for k, dist in enumerate(("ang",) if op in ("sin", "cos") else ("unit", "wide", "bits")):
        a, b, c, (na, nb, nc, no) = inputs(op, dist, 100 + k)
        want = amd(builtins_mod, op, a, b, c, no)
        got = np.zeros(N * no, np.float32)
        p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
        orc.lib.oracle_bi_eval(op.encode(), p(a), p(b), p(c), p(got), N)
        first_diff(f"cl_numerics.h {op} ({dist})", got, want, a)


@pytest.mark.gpu
@needs_gpu_ref
@pytest.mark.parametrize("op", sorted(HIP_OPS))
def test_hip_numerics_equal_amd_opencl_library(pkg, builtins_mod, op):
    from raytracing_amd.pyhost import mirt
    ctx = mirt.Context(0)
    try:
        for k, dist in enumerate(("ang", "bits") if op in ("sin", "cos") else ("unit", "wide", "bits")):
            a, b, c, (na, nb, nc, no) = inputs(op, dist, 200 + k)
            if dist == "bits" and op in ("sin", "cos"):
                a, b, c, _ = inputs("sqrt", "bits", 300 + k)     # every magnitude: the HIP side hands |x| >= 2^17 to ocml itself
                b = c = None
            want = amd(builtins_mod, op, a, b, c, no)
            if op in ("mad", "muladd"):
                got = ctx.debug_numerics(HIP_OPS[op], np.stack([a, b, c], axis=1).ravel())
            elif op == "clamp":
                got = np.empty(N, np.float32)
                for hi in (1.0, 255.0):
                    m = c == np.float32(hi)
                    got[m] = ctx.debug_numerics(25, a, np.full(N, hi, np.float32))[m]
            else:
                got = ctx.debug_numerics(HIP_OPS[op], a, b)
            first_diff(f"pt_numerics.hpp {op} ({dist})", got, want, a)
    finally:
        ctx.destroy()


@pytest.mark.gpu
@needs_gpu_ref
@pytest.mark.parametrize("name", [n for n in FULL_CASES if not n.endswith("_r1")] + ["cornell_320x240_r16"])
def test_reference_binary_on_the_device_equals_the_fixtures(name):
    """The reference's kernels (AMD OpenCL build) run on the MI355X through executeRender's sequence: every buffer after the primary
    segment and after the pass == the fixture, bit for bit."""
    from conftest import assert_state_equal
    fx, sc = load_fixture(name)
    k = G.GpuRefKernels()
    seeds = fx["seeds_in"] if "seeds_in" in fx else A.make_seeds(sc.total_rays)
    st = A.PassState(sc, seeds)
    ck = {}
    A.run_pass(k, sc, st, checkpoints=ck)
    k.release()
    assert np.array_equal(st.pixel, fx["pixel"]), "pixels"
    assert np.array_equal(canon(A.radiance_sums(st.acu, sc.rpp)), canon(fx["radiance"])), "radiance sums"
    if "f_acu" in fx:
        assert_state_equal(name + " primary", ck["primary"], fx, "p")
        assert_state_equal(name + " final", st.snapshot(), fx, "f")


@pytest.mark.gpu
@needs_gpu_ref
@pytest.mark.parametrize("name", ["basic_32x24_r4", "cornell_32x24_r4", "triangles_32x24_r4", "twoLights_32x24_r4", "threeLights_32x24_r1",
                                  "cornell_official_64x48_r1", "cornell_teapot3_32x24_r4", "own_flat_32x24_r4", "own_gems_48x36_r4", "own_studio_48x36_r4",
                                  "basic2_32x24_r4", "cornell_teapot_32x24_r4", "cornell_teapot2_32x24_r4"])
def test_every_scene_equals_the_reference_binary_at_depth_8(pkg, name):
    """Every scene the fixtures carry (the reference's A10 scenes incl. the grid-mesh ones, and ours) at 480x270 x 16 rays per pixel,
    depth 8, two progressive passes (2.07 M samples each): mirt_render_pass against the reference binary on the device, same seeds --
    every accumulator, seed and pixel."""
    from raytracing_amd.pyhost import mirt, render, scene
    fx, sc0 = load_fixture(name)
    ps = scene.PackedScene(dict(sc0.d)).resized(480, 270, 16)
    sc = A.Scene(ps.d)
    seeds = A.make_seeds(sc.total_rays, seed_base=11)
    k = G.GpuRefKernels()
    st = A.PassState(sc, seeds)
    ctx = mirt.Context(0)
    fr = render.FusedRenderer(ctx, ps, seeds=seeds)
    try:
        for p in range(2):
            A.run_pass(k, sc, st, bounces=8, init_acu=(p == 0))
            fr.execute_render(bounces=8)
        assert np.array_equal(canon(fr.acu.read(np.float32).reshape(-1, 4)), canon(st.acu)), "accumulators"
        assert np.array_equal(fr.seeds.read(np.int32), st.seeds), "seeds"
        assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), st.pixel), "pixels"
    finally:
        k.release()
        fr.release()
        ctx.destroy()


@pytest.mark.gpu
@needs_gpu_ref
def test_headline_frame_equals_the_reference_binary(pkg):
    """BASELINE's headline frame at FULL size -- cornell.xml 1920x1080 x 256 rays per pixel, depth 8: 530 841 600 samples -- rendered by
    the reference's own kernels (AMD OpenCL build) on the MI355X, 66 launches over 95 GB of rays / vertices / shadow rays that never leave
    the device, and by mirt_render_pass: every accumulator, every seed and every pixel equal, bit for bit."""
    from raytracing_amd.pyhost import mirt, render, scene
    base = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
    sc = A.Scene(base.d)
    n, npix = sc.total_rays, sc.width * sc.height
    assert n == 530841600
    ctx = mirt.Context(0)
    fr = render.FusedRenderer(ctx, base, want_radiance=False)           # seeds generated on the device (the closed form of make_seeds)
    seeds = fr.seeds.read(np.int32)
    fr.execute_render(bounces=8, fresh=True)
    ctx.finish()

    k = G.GpuRefKernels()

    class St:
        pass
    st = St()
    st.rays, st.pois, st.shadow = G.DevBuf(n * 48), G.DevBuf(n * 64), G.DevBuf(n * 48)
    st.acu, st.seeds, st.pixel = G.DevBuf(n * 16), G.DevBuf(n * 4), G.DevBuf(npix * 4)
    st.passes = 1
    st.seeds.upload(seeds)
    del seeds
    A.run_pass(k, sc, st, bounces=8)
    try:
        assert np.array_equal(st.pixel.download(np.uint8, npix * 4), fr.pixel.read(np.uint8)), "pixels"
        assert np.array_equal(st.seeds.download(np.int32, n), fr.seeds.read(np.int32)), "seeds"
        chunk = 1 << 26                                                # 64 M floats at a time: 8.5 GB of accumulators
        for off in range(0, 4 * n, chunk):
            m = min(chunk, 4 * n - off)
            want = np.empty(m, np.float32)
            G.chk(G.hip().hipMemcpy(want.ctypes.data_as(C.c_void_p), C.c_void_p(st.acu.ptr + 4 * off), 4 * m, 2), "D2H")
            got = fr.acu.read(np.float32, count=m, offset=4 * off)
            assert np.array_equal(canon(got), canon(want)), f"accumulators differ in floats [{off}, {off + m})"
    finally:
        for b in (st.rays, st.pois, st.shadow, st.acu, st.seeds, st.pixel):
            b.free()
        k.release()
        fr.release()
        ctx.destroy()


@pytest.mark.gpu
@needs_gpu_ref
@pytest.mark.parametrize("bounces", [5, 8])
def test_config5_lens_grid_band_equals_the_reference_binary(pkg, bounces):
    """BASELINE config 5's lens grid -- 3840x2160, 1024 rays per pixel = the 32 x 32 stratified thin-lens grid of initTrace
    (code.cl:482-509) -- pinned against the reference itself: its 2-D initTrace launched over global [3840, 8] (the first eight rows:
    31 457 280 rays, whose ids are exactly those of our row tile row0 = 0, nrows = 8; the camera still says 2160 rows) and its 1-D
    kernels over those rays, on the MI355X, against mirt_render_pass on the tile: every accumulator, seed and pixel.  (Below row 1092 the
    reference's own 32-bit `(cols*row+col)*rays_per_pixel` wraps: the top band is where "the reference's output" exists at this size.)"""
    from raytracing_amd.pyhost import mirt, render, scene
    base = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_3840x2160_r1024.json")).read())
    sc = A.Scene(base.d)
    assert (sc.width, sc.height, sc.rpp) == (3840, 2160, 1024) and sc.lens_rad > 0
    rows = 8
    n, npix = sc.width * rows * sc.rpp, sc.width * rows
    ctx = mirt.Context(0)
    fr = render.FusedRenderer(ctx, base, row0=0, nrows=rows, want_radiance=False)   # seeds generated on the device, global ids
    seeds = fr.seeds.read(np.int32)
    assert seeds.size == n
    fr.execute_render(bounces=bounces, fresh=True)
    ctx.finish()
    k = G.GpuRefKernels()

    class St:
        pass
    st = St()
    st.rays, st.pois, st.shadow = G.DevBuf(n * 48), G.DevBuf(n * 64), G.DevBuf(n * 48)
    st.acu, st.seeds, st.pixel = G.DevBuf(n * 16), G.DevBuf(n * 4), G.DevBuf(npix * 4)
    st.passes = 1
    st.seeds.upload(seeds)
    del seeds
    A.run_pass(k, sc, st, bounces=bounces, rows=rows)
    try:
        pix = fr.pixel.read(np.uint8)
        assert pix.reshape(-1, 4)[:, :3].max() > 0
        assert np.array_equal(st.pixel.download(np.uint8, npix * 4), pix), "pixels"
        assert np.array_equal(st.seeds.download(np.int32, n), fr.seeds.read(np.int32)), "seeds"
        assert np.array_equal(canon(st.acu.download(np.float32, 4 * n)), canon(fr.acu.read(np.float32))), "accumulators"
    finally:
        for b in (st.rays, st.pois, st.shadow, st.acu, st.seeds, st.pixel):
            b.free()
        k.release()
        fr.release()
        ctx.destroy()


# ---------------------------------------------------------------- CPU ----------------------------------------------------------------

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


@pytest.mark.skipif(not (os.path.exists(HSACO) and os.path.exists(READELF)), reason="needs the built code object and llvm-readelf (build container)")
@pytest.mark.parametrize("which", ["a10", "builtins"])
def test_launcher_layout_equals_the_code_objects_metadata(which):
    """oracle/ref_gpu.py packs kernel arguments by natural OpenCL C alignment; the code object's own metadata must say the same
    offsets and sizes for every explicit argument of every kernel."""
    path, table = (HSACO, G.KERNEL_ARGS) if which == "a10" else (BUILTINS, G.BUILTIN_ARGS)
    notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
    kernels = {}
    for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        args = [(int(o), int(sz), kind) for o, sz, kind in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+(?:\.type_name:\s+\S+\s+)?\.value_kind:\s+(\S+)", blk)]
        kernels[name] = [(o, sz) for o, sz, kind in args if not kind.startswith("hidden")]
    assert set(table) <= set(kernels), set(table) - set(kernels)
    for name, kinds in table.items():
        layout, _ = G.kernarg_layout(kinds)
        assert layout == kernels[name], f"{name}: launcher {layout} != metadata {kernels[name]}"
