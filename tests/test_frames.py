"""BASELINE configs 1-3: the single-frame kernels of Assign01 (one sphere), Assign04 (brute-force mesh) and Assign07
(3-D uniform grid, cell-parity shading).  Fixtures = the reference's own A01/A04/A07 host code + compiled code.cl
(oracle/gen/gen_golden_frame.py).  uchar4 frames and ray maxt must match exactly."""
import glob
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import a10_pass as A
import frame_pass as F
from conftest import GOLDEN, HOST, PAGE, ROOT, bits

CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "frame_*.npz")))
node = shutil.which("node")
REF = "/root/reference"
REF_TRI = {4: f"{REF}/Assign04-Triangle_Mesh/tri", 7: f"{REF}/Assign07-3D_uniform_grid_acceleration/tri"}
REF_MOL = f"{REF}/Assign07-3D_uniform_grid_acceleration/mol"


def fixture(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    return fx, json.loads(bytes(fx["frame_json"]).decode())


@pytest.mark.parametrize("name", CASES)
def test_oracle_frame_matches_compiled_reference(name):
    fx, d = fixture(name)
    px, rays = F.run_frame("oracle", F.Frame(d))
    assert np.array_equal(px, fx["pixel"])
    if rays is not None:
        assert np.array_equal(bits(rays["maxt"]), bits(fx["rays_maxt"])) and np.array_equal(bits(rays["mint"]), bits(fx["rays_mint"]))


def test_a01_fixture_is_the_expected_sphere():
    """Config 1: sphere c = (0,0,1), r = 0.5 seen from the origin along +z: t runs from 0.5 (centre) to sqrt(0.75) (silhouette),
    shade = (uchar)((1 - t) * 255) from 127 down to 34; about a fifth of the 2.66 x 2.0 window is covered."""
    fx, d = fixture("frame_a01_512x512")
    lit = fx["pixel"][:, 0]
    assert lit.max() == 127 and 33 <= lit[lit > 0].min() <= 36 and 0.18 < (lit > 0).mean() < 0.21
    assert (fx["pixel"][:, 3] == 255).all() and np.array_equal(fx["pixel"][:, 0], fx["pixel"][:, 1])


def mesh_path(d, name):
    if d["assign"] == 1:
        return "-"
    if "mol" in d:   # Assign07's molecule mode
        p = os.path.join(PAGE, "mol", d["mol"]) if "own" in name else os.path.join(REF_MOL, d["mol"])
        return p if os.path.exists(p) else None
    p = os.path.join(PAGE, "tri", d["mesh"]) if "own" in name else os.path.join(REF_TRI[d["assign"]], d["mesh"])
    return p if os.path.exists(p) else None


@pytest.mark.skipif(node is None, reason="node is not installed")
@pytest.mark.parametrize("name", CASES)
def test_js_host_packs_frames_like_the_reference_host(name):
    fx, d = fixture(name)
    mp = mesh_path(d, name)
    if mp is None:
        pytest.skip("reference mesh not present (GPU box)")
    r = subprocess.run([node, os.path.join(HOST, "cli.js"), "pack-frame", str(d["assign"]), mp, str(d["width"]), str(d["height"]),
                        str(d.get("n_slabs", 0))], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    got = json.loads(r.stdout)
    same = lambda a, b: np.array_equal(np.asarray(a, np.float64), np.asarray(b, np.float64), equal_nan=True)   # JSON null = undefined radius
    for k, v in d.items():
        if k in ("mesh", "mol"):
            continue
        if isinstance(v, list):
            assert same(v, got[k]), k
        elif isinstance(v, dict):   # "pdb": the molecule reader's own output (size, atomData, colorData, radiusData, min, max)
            for kk, vv in v.items():
                assert same(vv, got[k][kk]) if isinstance(vv, list) else vv == got[k][kk], f"{k}.{kk}"
        else:
            assert v == got[k], k


def test_program_dialects(pkg):
    """Kernel names collide across assignments; the OpenCL C text picks the set (mirt_program_dialect)."""
    from raytracing_amd.pyhost import mirt
    lib = mirt.lib()
    assert lib.mirt_program_dialect(b"__kernel void bouncePaths(){} __kernel void sceneRender(){} __kernel void initTrace(){}") == 10
    assert lib.mirt_program_dialect(b"__kernel void initTrace(){} __kernel void meshTrace(uint z_stride){}") == 7
    assert lib.mirt_program_dialect(b"__kernel void initTrace(){} __kernel void meshTrace(uint t_size){}") == 4
    assert lib.mirt_program_dialect(b"__kernel void raytrace(__global uchar4* p, float16 c){}") == 1
    assert lib.mirt_program_dialect(b"__kernel void raytrace(__global uchar4* p, __global float4* atoms){}") == 0      # A02
    assert lib.mirt_program_dialect(b"__kernel void meshTrace(uint n_slabs){}") == 0                                    # A05/A06
    assert lib.mirt_program_dialect(b"/* __kernel void bouncePaths(){} __kernel void sceneRender(){} */ __kernel void raytrace(float16 c){}") == 1
    if os.path.isdir(REF):
        got = [lib.mirt_program_dialect(open(f, "rb").read()) for f in sorted(glob.glob(f"{REF}/Assign*/code.cl"))]
        assert got == [1, 0, 0, 4, 0, 0, 7, 0, 0, 10]


@pytest.fixture(scope="module")
def ctx(pkg):
    from raytracing_amd.pyhost import mirt
    c = mirt.Context(0)
    yield c
    c.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_frame_matches_compiled_reference(ctx, pkg, name):
    from raytracing_amd.pyhost import render
    fx, d = fixture(name)
    px, rays = render.render_frame(ctx, render.FramePacked(d))
    assert np.array_equal(px, fx["pixel"])
    if rays is not None:
        r = rays.view(A.RAY_DT)
        assert np.array_equal(bits(r["maxt"]), bits(fx["rays_maxt"])) and np.array_equal(bits(r["mint"]), bits(fx["rays_mint"]))


@pytest.mark.gpu
@pytest.mark.skipif(node is None, reason="node is not installed")
@pytest.mark.parametrize("name", [c for c in CASES if "own" in c or "a01" in c])
def test_node_frame_matches_compiled_reference(tmp_path, name):
    """mesh.json -> JS host -> N-API -> HIP, for the meshes that exist on the GPU box (ours) and the mesh-less A01 job."""
    fx, d = fixture(name)
    out = str(tmp_path / "f.rgba")
    r = subprocess.run([node, os.path.join(HOST, "cli.js"), "frame", str(d["assign"]), mesh_path(d, name), str(d["width"]), str(d["height"]),
                        str(d.get("n_slabs", 0)), out], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(-1, 4), fx["pixel"])


def resized(d, w, h):
    """Same job at another frame size: Camera.set (A07 code.js:55-71) makes width = height * cols/rows; cols, rows ride in .sE/.sF."""
    d = dict(d, width=w, height=h)
    cam = list(d["cam"])
    cam[12] = float(np.float32(cam[13] * (w / h)))
    cam[14], cam[15] = float(w), float(h)
    d["cam"] = cam
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("name,size", [("frame_a04_parliament_96x64", (1024, 1024)),            # BASELINE config 2: 9144 triangles, brute force
                                       ("frame_a07_parliament_n16_160x120", (1920, 1080)),       # BASELINE config 3: same mesh, 16^3 grid
                                       ("frame_a04_own_icosphere_96x64", (1024, 1024)), ("frame_a07_own_octahedra_n3_96x64", (1920, 1080)),
                                       ("frame_a07_mol_3IZ4_n16_96x64", (1920, 1080)),           # molecule mode at scale: 8.9 k atoms, 16^3 grid
                                       ("frame_a07_own_mol_lattice_n6_96x64", (1920, 1080))])
def test_full_size_frames_against_oracle(ctx, pkg, name, size):
    _full_size(ctx, name, size)


def regrid(a07_job, a04_job, n):
    """The Assign07 job of the same mesh at another n_slabs: the reference host's binning (tests/test_grid_build.py's restatement,
    itself checked against the reference host's grids the fixtures carry) over the mesh's triangles in input order."""
    from test_grid_build import expected_grid, unique_triangles
    tri = unique_triangles(a04_job)
    nor = np.asarray(a04_job["normal"], np.float32).reshape(-1, 12)
    b = np.asarray(a07_job["bounds"], np.float64)
    off, order = expected_grid(1, tri, [b[0], b[1], b[2], b[4], b[5], b[6]], n)
    pos = np.zeros((len(order), 3, 4), np.float32)
    pos[:, :, :3] = tri[order].astype(np.float32).reshape(-1, 3, 3)
    return dict(a07_job, n_slabs=n, slab_size=off.tolist(), pos=pos.ravel().tolist(), normal=nor[order].ravel().tolist(),
                mindex=np.asarray(a04_job["mindex"], np.uint32)[order].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 32])
def test_parliament_grid_variants_at_full_size(ctx, pkg, n):
    """BASELINE config 3's other grid resolutions (n_slabs in {2 = the page's default, 16, 32}) at 1920x1080."""
    from raytracing_amd.pyhost import render
    _, g = fixture("frame_a07_parliament_n16_160x120")
    _, flat = fixture("frame_a04_parliament_96x64")
    d = resized(regrid(g, flat, n), 1920, 1080)
    px, _ = render.render_frame(ctx, render.FramePacked(d))
    want, _ = F.run_frame("oracle", F.Frame(d))
    assert np.array_equal(px, want) and (px[:, :3].max(axis=1) > 0).mean() > 0.03


@pytest.mark.gpu
def test_teapot_brute_force_at_full_size(ctx, pkg):
    """BASELINE config 2 on the reference's other mesh: teapot.json (992 triangles), 1024 x 1024, every pixel against every triangle."""
    _full_size(ctx, "frame_a04_teapot_160x120", (1024, 1024))


def _full_size(ctx, name, size):
    """BASELINE configs 2 and 3 at their full sizes, on the reference's house_of_parliament mesh (its packed buffers travel inside
    the fixture) and on ours: HIP frame == multithreaded CPU oracle, every pixel."""
    from raytracing_amd.pyhost import render
    fx, d = fixture(name)
    d = resized(d, *size)
    ctx.timer_start()
    px, _ = render.render_frame(ctx, render.FramePacked(d))
    ms = ctx.timer_stop_ms()
    want, _ = F.run_frame("oracle", F.Frame(d))
    assert np.array_equal(px, want)
    print(f"\n{name} at {size[0]}x{size[1]}: {ms:.2f} ms incl. uploads")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "a07_gfx950.hsaco")), reason="oracle/_ref/*.hsaco not built (make -C oracle ref_gpu, build container)")
@pytest.mark.parametrize("name,size", [("frame_a01_512x512", None), ("frame_a04_parliament_96x64", (1024, 1024)), ("frame_a04_teapot_160x120", (1024, 1024)),
                                       ("frame_a07_parliament_n16_160x120", (1920, 1080)), ("frame_a07_teapot_n2_160x120", (1920, 1080)),
                                       ("frame_a07_mol_3IZ4_n16_96x64", (1920, 1080)), ("frame_a07_mol_c60_n4_160x120", (1920, 1080)),
                                       ("frame_a07_own_terrain_n5_96x64", (1920, 1080)),
                                       ("frame_a04_house_160x120", (1024, 1024)), ("frame_a07_house_n2_160x120", (1920, 1080)), ("frame_a07_house_n8_160x120", (1920, 1080))])
def test_full_size_frames_equal_the_reference_binaries(ctx, pkg, name, size):
    """BASELINE configs 1-3 (and the molecule mode) at full size against the REFERENCE'S OWN kernels: Assign01 / 04 / 07 code.cl as AMD's
    OpenCL toolchain builds them for gfx950, run on the device (oracle/frame_pass.run_frame_gpu): every pixel, and every ray's maxt."""
    from raytracing_amd.pyhost import render
    fx, d = fixture(name)
    if size:
        d = resized(d, *size)
    want_px, want_rays = F.run_frame_gpu(F.Frame(d))
    px, rays = render.render_frame(ctx, render.FramePacked(d))
    assert np.array_equal(px, want_px)
    if want_rays is not None and rays is not None:
        got = np.ascontiguousarray(rays).view(A.RAY_DT)
        assert np.array_equal(bits(got["maxt"]), bits(want_rays["maxt"]))
    assert (px[:, :3].max(axis=1) > 0).mean() > 0.02


def _random_mesh_job(base04, base07, seed, n):
    """A random triangle soup inside the parliament fixture's box (its camera then sees it): Assign04 job, or the Assign07 job at n_slabs = n
    binned by the restatement of the reference host's splitMeshData.  Sizes from specks to slivers as long as the box; a few degenerate ones."""
    from test_grid_build import expected_grid
    rng = np.random.default_rng(seed)
    b = np.asarray(base04["bounds"], np.float64)
    lo, hi = b[:3], b[4:7]
    T = int(rng.choice([5, 120, 900, 2500]))
    c = lo + rng.uniform(0.05, 0.95, size=(T, 1, 3)) * (hi - lo)
    scale = (hi - lo) * rng.choice([0.003, 0.02, 0.2, 1.0], size=(T, 1, 1))
    v = (c + rng.uniform(-0.5, 0.5, size=(T, 3, 3)) * scale).astype(np.float32)
    v[::97, 2] = v[::97, 1]                                    # degenerate: two equal vertices
    v = np.clip(v, lo.astype(np.float32), hi.astype(np.float32))
    nrm = rng.normal(size=(T, 3, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=2, keepdims=True)).astype(np.float32)
    ncol = len(base04["mcolor"]) // 4
    mindex = rng.integers(0, ncol, size=T).astype(np.uint32)

    def pack(order):
        pos = np.zeros((len(order), 3, 4), np.float32)
        nor = np.zeros((len(order), 3, 4), np.float32)
        pos[:, :, :3] = v[order]
        nor[:, :, :3] = nrm[order]
        return pos.ravel().tolist(), nor.ravel().tolist()
    if n == 0:
        pos, nor = pack(np.arange(T))
        return dict(base04, t_size=T, pos=pos, normal=nor, mindex=mindex.tolist())
    off, order = expected_grid(1, v.reshape(T, 9).astype(np.float64), [b[0], b[1], b[2], b[4], b[5], b[6]], n)
    pos, nor = pack(order)
    return dict(base07, t_size=T, n_slabs=n, slab_size=off.tolist(), pos=pos, normal=nor, mindex=mindex[order].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(s, n) for s in range(6) for n in (0, 1, 2, 5)])
def test_random_meshes_frames_match_oracle(ctx, pkg, seed, n):
    """Random triangle soups through the Assign04 brute-force kernel (n = 0) and the Assign07 grid kernel at 1, 2 and 5 cells per axis: the
    per-group bounding spheres (pt_trace.hpp group_missed) must never skip a triangle the reference's test accepts -- every pixel and every
    ray's maxt against the CPU oracle."""
    from raytracing_amd.pyhost import render
    _, a04 = fixture("frame_a04_parliament_96x64")
    _, a07 = fixture("frame_a07_parliament_n16_160x120")
    d = resized(_random_mesh_job(a04, a07, 100 + seed, n), 320, 200)
    px, rays = render.render_frame(ctx, render.FramePacked(d))
    want, wrays = F.run_frame("oracle", F.Frame(d))
    assert np.array_equal(px, want)
    assert (px[:, :3].max(axis=1) > 0).mean() > 0.004


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 16, 32])
def test_device_regrid_equals_the_host_binning(ctx, pkg, n):
    """pyhost render.frame_regrid (bench.py's configs 2 / 3 record bins house_of_parliament at n = 2 and 32 with it: mirt_grid_build + gathers)
    against the restatement of the reference host's splitMeshData -- and, at n = 16, against the reference host's own grid the fixture carries."""
    from raytracing_amd.pyhost import render
    _, g = fixture("frame_a07_parliament_n16_160x120")
    _, flat = fixture("frame_a04_parliament_96x64")
    got = render.frame_regrid(ctx, g, flat, n)
    want = g if n == 16 else regrid(g, flat, n)
    assert got["n_slabs"] == n and got["slab_size"] == [int(x) for x in want["slab_size"]]
    for key in ("pos", "normal"):
        assert np.array_equal(bits(np.asarray(got[key], np.float32)), bits(np.asarray(want[key], np.float32))), key
    assert got["mindex"] == [int(x) for x in want["mindex"]]
    t = {}
    px, _ = render.render_frame(ctx, render.FramePacked(render.frame_resized(got, 320, 180)), timing=t)
    px2, _ = render.render_frame(ctx, render.FramePacked(resized(want, 320, 180)))
    assert np.array_equal(px, px2) and t["trace_ms"] > 0
