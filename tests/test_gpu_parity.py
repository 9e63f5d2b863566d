"""GPU: the HIP path, called through the C ABI, against (a) the golden fixtures = outputs of the
compiled reference kernels and (b) the CPU oracle on the same seeded inputs.  Bit-exact: integer hit
ids (matId), seeds, every fp32 buffer.  Both kernel families are covered: the fourteen reference-shaped
kernels enqueued in executeRender's order, and the fused one-launch pass."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import a10_pass as A
from conftest import FULL_CASES, assert_state_equal, bits, load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    from raytracing_amd.pyhost import mirt
    c = mirt.Context(0)
    yield c
    c.destroy()


def snapshot(gr):
    return {"rays": gr.read("rays").view(A.RAY_DT), "shadow": gr.read("shadow").view(A.RAY_DT),
            "pois": gr.read("pois").view(A.POI_DT), "acu": gr.read("acu").reshape(-1, 4), "seeds": gr.read("seeds")}


def test_struct_sizes_reported_by_device(ctx, pkg):
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("basic_32x24_r4")
    gr = render.GranularRenderer(ctx, sc, seeds=fx["seeds_in"])
    assert (gr.ray_size, gr.poi_size) == (48, 64)
    gr.release()


@pytest.mark.parametrize("name", FULL_CASES)
def test_granular_kernels_match_compiled_reference(ctx, pkg, name):
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture(name)
    gr = render.GranularRenderer(ctx, sc, seeds=fx["seeds_in"])
    primary = {}
    gr.execute_render(on_primary=lambda g: primary.update(snapshot(g)))
    assert_state_equal(name + ":primary", primary, fx, "p")
    assert_state_equal(name + ":final", snapshot(gr), fx, "f")
    assert np.array_equal(gr.read("pixel").reshape(-1, 4), fx["pixel"])
    gr.release()


@pytest.mark.parametrize("name", FULL_CASES)
def test_fused_pass_matches_compiled_reference(ctx, pkg, name):
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture(name)
    fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"])
    fr.execute_render()
    assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(fx["f_acu"])), "acu"
    assert np.array_equal(fr.seeds.read(np.int32), fx["f_seeds"]), "seeds"
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), fx["pixel"]), "pixel"
    assert np.array_equal(bits(fr.radiance.read(np.float32).reshape(-1, 4)), bits(fx["radiance"])), "radiance"
    fr.release()


@pytest.mark.parametrize("exact_only", [False, True], ids=["optimistic", "exact_only"])
@pytest.mark.parametrize("name", FULL_CASES)
def test_both_fused_modes_match_compiled_reference(ctx, pkg, name, exact_only):
    """The default two-kernel pass (3-operation exact divisions + exact re-run of the samples outside the guard window) and
    the single exact kernel (mirt_ctx_set_exact_only) must give the same bits; samples with NaN rays (odd lens grid) are
    among the deferred ones."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture(name)
    ctx.set_exact_only(exact_only)
    try:
        fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"])
        fr.execute_render()
        deferred = ctx.pass_deferred()
        assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(fx["f_acu"])), "acu"
        assert np.array_equal(fr.seeds.read(np.int32), fx["f_seeds"]), "seeds"
        assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), fx["pixel"]), "pixel"
        fr.release()
    finally:
        ctx.set_exact_only(False)
    if exact_only:
        assert deferred == 0
    elif name == "cornell_16x12_r9":
        assert deferred >= sc.total_rays // 9          # every centre-of-lens sample is a NaN ray
    assert deferred <= sc.total_rays


def test_fused_large_case_against_fixture_digests(ctx, pkg):
    """cornell.xml 320x240 x 16 rays per pixel: pixel / radiance exact, per-ray acu + seeds by SHA-256."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("cornell_320x240_r16")
    fr = render.FusedRenderer(ctx, sc)            # seeds generated on the device: same closed form
    fr.execute_render()
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), fx["pixel"])
    assert np.array_equal(bits(fr.radiance.read(np.float32).reshape(-1, 4)), bits(fx["radiance"]))
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha(fr.acu.read(np.float32)), fx["sha_acu"])
    assert np.array_equal(sha(fr.seeds.read(np.int32)), fx["sha_seeds"])
    fr.release()


def test_device_seed_fill_equals_host_formula(ctx):
    b = ctx.buffer(4 * 100000)
    ctx.seed_fill(b, 12345, 100000, 0)
    assert np.array_equal(b.read(np.int32), A.make_seeds(100000, first=12345))
    ctx.seed_fill(b, 0, 100000, 77)
    assert np.array_equal(b.read(np.int32), A.make_seeds(100000, seed_base=77))
    b.release()


def test_row_tiles_compose_to_the_full_frame(ctx, pkg):
    """Multi-GPU sharding contract: rendering rows [0,h/2) and [h/2,h) separately gives the same bytes as one frame."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("cornell_teapot3_32x24_r4")
    tiles = []
    for row0, nrows in ((0, 10), (10, 14)):
        fr = render.FusedRenderer(ctx, sc, row0=row0, nrows=nrows)
        fr.execute_render()
        tiles.append((fr.pixel.read(np.uint8).reshape(-1, 4), fr.radiance.read(np.float32).reshape(-1, 4)))
        fr.release()
    assert np.array_equal(np.concatenate([t[0] for t in tiles]), fx["pixel"])
    assert np.array_equal(bits(np.concatenate([t[1] for t in tiles])), bits(fx["radiance"]))


def test_progressive_passes_match_oracle(ctx, pkg):
    """Three passes into the same accumulator (A10 code.js:1850-1853), depth 8 variant included."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("twoLights_32x24_r4")
    orc = A.load_oracle()
    for bounces in (5, 8):
        st = A.PassState(sc, fx["seeds_in"])
        fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"])
        for p in range(3):
            A.run_pass(orc, sc, st, bounces=bounces, init_acu=(p == 0))
            fr.execute_render(bounces=bounces)
        assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu))
        assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), st.pixel)
        fr.release()


@pytest.mark.parametrize("rpp", [2, 5])
def test_granular_pass_with_a_non_square_ray_count(ctx, pkg, rpp):
    """The host always asks for k x k rays per pixel (A10 code.js:540), but the kernels take any count: initTrace's k x k loops write
    the first k*k rays of a pixel and leave the rest as the buffer held them (code.cl:479-512) while resetting ALL rpp vertices.
    With zeroed buffers on both sides the tail rays are dead (mint == maxt == 0) and the whole pass is defined: kernel-by-kernel HIP
    == oracle.  The fused pass refuses such a count."""
    from raytracing_amd.pyhost import mirt, render
    fx, sc0 = load_fixture("cornell_32x24_r4")
    sc = A.Scene(dict(sc0.d, rays_per_pixel=rpp))
    seeds = A.make_seeds(sc.total_rays)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    A.run_pass(orc, sc, st)
    gr = render.GranularRenderer(ctx, sc, seeds=seeds)
    for name in ("rays", "pois", "shadow"):
        ctx.zero(gr.b[name])
    gr.execute_render()
    got = snapshot(gr)
    assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds)
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"])
    for f in ("mint", "maxt"):
        assert np.array_equal(bits(got["rays"][f]), bits(st.rays[f]))
    assert np.array_equal(gr.read("pixel").reshape(-1, 4), st.pixel)
    gr.release()
    with pytest.raises(mirt.MirtError) as e:
        render.FusedRenderer(ctx, sc, seeds=seeds).execute_render()
    assert e.value.code == -1 and "not a square" in str(e.value)


@pytest.mark.parametrize("name", ["cornell_64x48_r1", "cornell_teapot3_32x24_r4"])
def test_graph_replayed_passes_match_oracle(ctx, pkg, name):
    """Five progressive passes of the kernel-by-kernel path; from the third pass on the pass body is a HIP graph replay
    (mirt_capture_begin / end / mirt_graph_launch).  rpp 1 has the serial lens pre-pass inside initTrace; the teapot scene has per-mesh
    argument changes between launches -- both are recorded by value."""
    from raytracing_amd.pyhost import mirt, render
    fx, sc = load_fixture(name)
    orc = A.load_oracle()
    st = A.PassState(sc, fx["seeds_in"])
    gr = render.GranularRenderer(ctx, sc, seeds=fx["seeds_in"])
    for p in range(5):
        A.run_pass(orc, sc, st, init_acu=(p == 0))
        gr.execute_render(use_graph=True)
        got = snapshot(gr)
        assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds), f"pass {p}"
        assert np.array_equal(gr.read("pixel").reshape(-1, 4), st.pixel), f"pixel, pass {p}"
    assert gr._graph is not None
    # what cannot be recorded says so, and the recording survives it
    ctx.capture_begin()
    with pytest.raises(mirt.MirtError) as e:
        ctx.finish()
    assert e.value.code == -1 and "capture" in str(e.value)
    g = ctx.capture_end()
    ctx.graph_release(g)
    gr.release()


def _variant(sc0, **kw):
    """The same packed scene with fields replaced; width / height changes re-pack the camera the way Camera.lookAt does."""
    d = dict(sc0.d)
    w, h = kw.pop("width", d["width"]), kw.pop("height", d["height"])
    cam = list(d["cam"])
    cam[12] = float(np.float32(cam[13] * (w / h)))
    cam[14], cam[15] = float(w), float(h)
    d.update(cam=cam, width=w, height=h, **kw)
    return A.Scene(d)


DEGENERATE = {
    "no_lights": lambda sc: _variant(sc, lights=[]),                                            # nothing is ever shaded; paths still bounce
    "no_geometry": lambda sc: _variant(sc, n_spheres=0, n_triangles=0, meshes=[]),              # every ray misses everything
    "spheres_only_ragged_13x7": lambda sc: _variant(sc, n_triangles=0, meshes=[], width=13, height=7),   # image not a multiple of any block
    "one_pixel": lambda sc: _variant(sc, width=1, height=1),
    "one_column_rpp1": lambda sc: _variant(sc, width=1, height=9, rays_per_pixel=1),            # the seeds[col] stream feeds every row
    "tall_sliver_2x67": lambda sc: _variant(sc, width=2, height=67),
    # a triangle that reaches far outside the box its set declares (the box is the caller's word): exact kernel, same answer
    "vertex_beyond_2p21": lambda sc: _variant(sc, meshes=[], t_pos=[5.0e6 if i == 4 else v for i, v in enumerate(sc.d["t_pos"])]),
    # an inverted box (min > max on one axis) is a miss for every ray in the reference; the optimistic kernel must not see it
    "inverted_sphere_box": lambda sc: _variant(sc, meshes=[], sphere_bounds=[b if i != 0 and i != 4 else sc.d["sphere_bounds"][4 - i] for i, b in enumerate(sc.d["sphere_bounds"])]),
}


@pytest.mark.parametrize("case", sorted(DEGENERATE))
def test_degenerate_scenes_match_oracle(ctx, pkg, case):
    """Empty and ragged inputs: no lights, no primitives, image sizes that fill no block / wave / 8x8 work-group, a single pixel, a single
    column at rpp 1.  Both HIP paths against the oracle on the same seeds."""
    from raytracing_amd.pyhost import render
    fx, sc0 = load_fixture("cornell_teapot3_32x24_r4")
    sc = DEGENERATE[case](sc0)
    seeds = A.make_seeds(sc.total_rays)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    A.run_pass(orc, sc, st)
    gr = render.GranularRenderer(ctx, sc, seeds=seeds)
    gr.execute_render()
    got = snapshot(gr)
    assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds), "granular"
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"])
    assert np.array_equal(gr.read("pixel").reshape(-1, 4), st.pixel)
    gr.release()
    fr = render.FusedRenderer(ctx, sc, seeds=seeds)
    fr.execute_render()
    if case in ("vertex_beyond_2p21", "inverted_sphere_box"):
        assert ctx.pass_deferred() == 0            # the optimistic kernel was not used at all
    assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu)), "fused"
    assert np.array_equal(fr.seeds.read(np.int32), st.seeds)
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), st.pixel)
    fr.release()


MEDIUM = ["basic_32x24_r4", "cornell_32x24_r4", "triangles_32x24_r4", "twoLights_32x24_r4", "threeLights_32x24_r1", "cornell_official_64x48_r1",
          "cornell_teapot3_32x24_r4", "own_flat_32x24_r4", "own_gems_48x36_r4", "own_studio_48x36_r4",
          "basic2_32x24_r4", "cornell_teapot_32x24_r4", "cornell_teapot2_32x24_r4"]


@pytest.mark.parametrize("name", MEDIUM)
def test_medium_frames_match_oracle(ctx, pkg, name):
    """Every scene the fixtures carry (the reference's A10 scenes and ours), re-sized to 240x135 at 16 rays per pixel and depth 8
    (518 k samples, two progressive passes): the fused pass against the CPU oracle on the same seeds, every accumulator and seed."""
    from raytracing_amd.pyhost import render
    fx, sc0 = load_fixture(name)
    sc = _variant(sc0, width=240, height=135, rays_per_pixel=16)
    seeds = A.make_seeds(sc.total_rays, seed_base=5)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    fr = render.FusedRenderer(ctx, sc, seeds=seeds)
    for p in range(2):
        A.run_pass(orc, sc, st, bounces=8, init_acu=(p == 0))
        fr.execute_render(bounces=8)
    assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu))
    assert np.array_equal(fr.seeds.read(np.int32), st.seeds)
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), st.pixel)
    fr.release()


def _regrid_mesh(m, n):
    """A packed A10 mesh (cell-sorted, duplicated triangles) re-binned at another grid resolution by the restatement of the reference
    host's splitMeshData (tests/test_grid_build.py).  The triangle set is the mesh's distinct (position, normal) records."""
    from test_grid_build import expected_grid
    pos = np.asarray(m["pos"], np.float32).reshape(-1, 12)
    nor = np.asarray(m["normal"], np.float32).reshape(-1, 12)
    _, first = np.unique(np.concatenate([pos, nor], axis=1), axis=0, return_index=True)
    first.sort()
    pos, nor = pos[first], nor[first]
    tri = pos.reshape(-1, 3, 4)[:, :, :3].reshape(-1, 9).astype(np.float64)
    b = np.asarray(m["bounds"], np.float64)
    off, order = expected_grid(1, tri, [b[0], b[1], b[2], b[4], b[5], b[6]], n)
    return dict(m, nslabs=n, box=off.tolist(), pos=pos[order].ravel().tolist(), normal=nor[order].ravel().tolist())


def _sparse_grid(tri, bounds6, n):
    """expected_grid (tests/test_grid_build.py) without its n^3 Python lists: the same cell ranges per triangle, a stable sort by cell."""
    tri = np.asarray(tri, np.float64).reshape(-1, 3, 3)
    bmin = np.asarray(bounds6[:3], np.float64)
    w = (np.asarray(bounds6[3:], np.float64) - bmin) / n
    a = np.maximum(np.floor((tri.min(axis=1) - bmin) / w), 0).astype(np.int64)
    b = np.minimum(np.floor((tri.max(axis=1) - bmin) / w), n - 1).astype(np.int64)
    cells, ids = [], []
    for i in range(len(tri)):
        if (a[i] > b[i]).any():
            continue
        z, y, x = np.meshgrid(*(np.arange(a[i][k], b[i][k] + 1) for k in (2, 1, 0)), indexing="ij")
        c = ((z * n + y) * n + x).ravel()
        cells.append(c)
        ids.append(np.full(c.size, i, np.int64))
    cells, ids = np.concatenate(cells), np.concatenate(ids)
    order = np.lexsort((ids, cells))
    off = np.zeros(n ** 3 + 1, np.uint32)
    off[1:] = np.cumsum(np.bincount(cells, minlength=n ** 3))
    return off, ids[order].astype(np.uint32)


def test_small_triangles_in_a_300_cube_grid(ctx, pkg):
    """Slab indices beyond eight bits: 300 small triangles in a 300^3 grid (27 M cells, a 108 MB offset table read from memory) in place of
    cornell_teapot3's box.  The shared-test walk packs its three slab indices ten bits each and carries the cell index along
    (pt_trace_coop.hpp): rays cross up to 900 cells here.  Fused pass (both modes) and kernel-by-kernel path against the CPU oracle."""
    from raytracing_amd.pyhost import render
    fx, sc0 = load_fixture("cornell_teapot3_32x24_r4")
    m = sc0.d["meshes"][1]
    b = np.asarray(m["bounds"], np.float64)
    rng = np.random.default_rng(300)
    c = b[:3] + (b[4:7] - b[:3]) * rng.uniform(0.05, 0.95, (300, 1, 3))
    tri = (c + rng.uniform(-0.05, 0.05, (300, 3, 3))).astype(np.float32)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float64)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    pos4 = np.concatenate([tri, np.ones((300, 3, 1), np.float32)], axis=2).reshape(300, 12)
    nor4 = np.concatenate([np.repeat(nrm[:, None, :], 3, axis=1), np.zeros((300, 3, 1), np.float32)], axis=2).reshape(300, 12)
    n = 300
    off, order = _sparse_grid(tri.astype(np.float64).reshape(-1, 9), [b[0], b[1], b[2], b[4], b[5], b[6]], n)
    assert 100_000 < order.size < 3_000_000
    mesh = dict(m, nslabs=n, box=off, pos=pos4[order].ravel(), normal=nor4[order].ravel())
    sc = _variant(sc0, width=96, height=54, rays_per_pixel=4, meshes=[sc0.d["meshes"][0], mesh])
    seeds = A.make_seeds(sc.total_rays, seed_base=n)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    A.run_pass(orc, sc, st)
    assert (st.pois["matId"] == m["matid"]).mean() > 0.005   # some paths end on the small triangles
    for exact_only in (False, True):
        ctx.set_exact_only(exact_only)
        fr = render.FusedRenderer(ctx, sc, seeds=seeds)
        fr.execute_render()
        assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu)), f"fused, exact_only={exact_only}"
        assert np.array_equal(fr.seeds.read(np.int32), st.seeds)
        fr.release()
    ctx.set_exact_only(False)
    gr = render.GranularRenderer(ctx, sc, seeds=seeds)
    gr.execute_render()
    got = snapshot(gr)
    assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds), "granular"
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"])
    gr.release()


@pytest.mark.parametrize("n", [2, 17, 24])
def test_teapot_at_other_grid_resolutions(ctx, pkg, n):
    """cornell_teapot3 with its teapot re-binned: n = 2 puts hundreds of triangles in a cell (the shared-test walk lays one cell out over
    many 64-pair rounds), n = 17 and 24 have cell-offset tables that do not fit the LDS budget (kLdsOffWords: the fused pass then reads
    every table from memory, k_fusedPass<*, 2>).  Fused pass (both modes) and kernel-by-kernel path against the CPU oracle."""
    from raytracing_amd.pyhost import render
    fx, sc0 = load_fixture("cornell_teapot3_32x24_r4")
    meshes = [_regrid_mesh(sc0.d["meshes"][0], n)] + list(sc0.d["meshes"][1:])
    sc = _variant(sc0, width=96, height=54, rays_per_pixel=4, meshes=meshes)
    seeds = A.make_seeds(sc.total_rays, seed_base=n)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    A.run_pass(orc, sc, st)
    assert (st.pois["matId"] == sc.d["meshes"][0]["matid"]).mean() > 0.02   # the teapot is in the picture
    for exact_only in (False, True):
        ctx.set_exact_only(exact_only)
        fr = render.FusedRenderer(ctx, sc, seeds=seeds)
        fr.execute_render()
        assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu)), f"fused, exact_only={exact_only}"
        assert np.array_equal(fr.seeds.read(np.int32), st.seeds)
        assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), st.pixel)
        fr.release()
    ctx.set_exact_only(False)
    gr = render.GranularRenderer(ctx, sc, seeds=seeds)
    gr.execute_render()
    got = snapshot(gr)
    assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds), "granular"
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"])
    gr.release()


def _regrid_loose_sets(d, n):
    """The scene's loose spheres and triangles binned at n_slabs = n (the reference host uses 1 for them, A10 code.js:399, but its kernels
    take any): cell lists by the restatement of splitSphereData / splitTriangleData over the packed primitives."""
    from test_grid_build import expected_grid
    out = {"n_slabs": n}
    if d["n_spheres"]:
        sph = np.asarray(d["spheres"], np.float32).reshape(-1, 4)
        b = np.asarray(d["sphere_bounds"], np.float64)
        cr = np.concatenate([sph[:, :3].astype(np.float64), np.sqrt(sph[:, 3:4].astype(np.float64))], axis=1)   # packed (c, r^2)
        off, order = expected_grid(0, cr, [b[0], b[1], b[2], b[4], b[5], b[6]], n)
        out.update(spheres=sph[order].ravel().tolist(), s_matid=np.asarray(d["s_matid"])[order].tolist(), s_box=off.tolist(), n_spheres=len(order))
    if d["n_triangles"]:
        pos = np.asarray(d["t_pos"], np.float32).reshape(-1, 12)
        nor = np.asarray(d["t_normal"], np.float32).reshape(-1, 12)
        b = np.asarray(d["triangle_bounds"], np.float64)
        tri = pos.reshape(-1, 3, 4)[:, :, :3].reshape(-1, 9).astype(np.float64)
        off, order = expected_grid(1, tri, [b[0], b[1], b[2], b[4], b[5], b[6]], n)
        out.update(t_pos=pos[order].ravel().tolist(), t_normal=nor[order].ravel().tolist(), t_matid=np.asarray(d["t_matid"])[order].tolist(),
                   t_box=off.tolist(), n_triangles=len(order))
    return out


@pytest.mark.parametrize("name,n", [("basic_32x24_r4", 3), ("triangles_32x24_r4", 2), ("cornell_32x24_r4", 4), ("own_gems_48x36_r4", 3)])
def test_loose_spheres_and_triangles_in_grids(ctx, pkg, name, n):
    """sphereTrace / triangleTrace and their shadow kernels with n_slabs > 1: sphere sets walk the per-lane DDA, loose-triangle sets
    (per-primitive material ids) the shared-test walk, next to meshes with their own grids where the scene has them."""
    from raytracing_amd.pyhost import render
    fx, sc0 = load_fixture(name)
    sc = _variant(sc0, width=80, height=45, rays_per_pixel=4, **_regrid_loose_sets(sc0.d, n))
    seeds = A.make_seeds(sc.total_rays, seed_base=7 * n)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    A.run_pass(orc, sc, st)
    assert (st.pois["matId"] >= 0).mean() > 0.1
    for exact_only in (False, True):
        ctx.set_exact_only(exact_only)
        fr = render.FusedRenderer(ctx, sc, seeds=seeds)
        fr.execute_render()
        assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu)), f"fused, exact_only={exact_only}"
        assert np.array_equal(fr.seeds.read(np.int32), st.seeds)
        fr.release()
    ctx.set_exact_only(False)
    gr = render.GranularRenderer(ctx, sc, seeds=seeds)
    gr.execute_render()
    got = snapshot(gr)
    assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds), "granular"
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"])
    for f in ("mint", "maxt"):
        assert np.array_equal(bits(got["shadow"][f]), bits(st.shadow[f])), f"shadow.{f}"
    gr.release()


def _mesh_with_bounds(m, f):
    b = list(m["bounds"])
    f(b)
    return dict(m, bounds=b)


def _flatten_z(b):
    b[6] = b[2]                      # zero-width box on z: the slab width is 0, every quotient by it is inf or NaN


def _invert_x(b):
    b[0], b[4] = b[4], b[0]          # min > max: a miss for every ray (code.cl:301-333)


def _inflate(b):
    for k in range(3):
        c, h = 0.5 * (b[k] + b[4 + k]), 0.5 * (b[4 + k] - b[k])
        b[k], b[4 + k] = float(np.float32(c - 7.0 * h)), float(np.float32(c + 7.0 * h))   # the lists no longer match the cells: windows reject most hits


GRID_DEGENERATE = {
    "nan_rays_rpp9": lambda sc: _variant(sc, width=40, height=24, rays_per_pixel=9),                       # the centre sample of a 3 x 3 lens grid is NaN
    "teapot_box_zero_width": lambda sc: _variant(sc, meshes=[_mesh_with_bounds(sc.d["meshes"][0], _flatten_z)] + list(sc.d["meshes"][1:])),
    "room_box_zero_width": lambda sc: _variant(sc, meshes=[sc.d["meshes"][0], _mesh_with_bounds(sc.d["meshes"][1], _flatten_z)]),
    "teapot_box_inverted": lambda sc: _variant(sc, meshes=[_mesh_with_bounds(sc.d["meshes"][0], _invert_x)] + list(sc.d["meshes"][1:])),
    "room_box_inflated": lambda sc: _variant(sc, meshes=[sc.d["meshes"][0], _mesh_with_bounds(sc.d["meshes"][1], _inflate)]),
    "one_wave_row_65x1": lambda sc: _variant(sc, width=65, height=1, rays_per_pixel=4),                    # 260 rays: one full block and four lanes of the next
}


@pytest.mark.parametrize("case", sorted(GRID_DEGENERATE))
def test_degenerate_grids_match_oracle(ctx, pkg, case):
    """Grid sets whose declared box is the caller's word and a bad one (zero width: the slab width divides by zero; inverted; far larger than
    the geometry, so that cell lists and cells no longer agree), NaN rays through the grids, and a tile that leaves most of a block without
    samples.  The shared-test walk, the exact kernel and the kernel-by-kernel path against the CPU oracle."""
    from raytracing_amd.pyhost import render
    fx, sc0 = load_fixture("cornell_teapot3_32x24_r4")
    sc = GRID_DEGENERATE[case](sc0)
    seeds = A.make_seeds(sc.total_rays, seed_base=11)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    A.run_pass(orc, sc, st)
    for exact_only in (False, True):
        ctx.set_exact_only(exact_only)
        fr = render.FusedRenderer(ctx, sc, seeds=seeds)
        fr.execute_render()
        assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu)), f"fused, exact_only={exact_only}"
        assert np.array_equal(fr.seeds.read(np.int32), st.seeds)
        assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), st.pixel)
        fr.release()
    ctx.set_exact_only(False)
    gr = render.GranularRenderer(ctx, sc, seeds=seeds)
    gr.execute_render()
    got = snapshot(gr)
    assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds), "granular"
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"])
    gr.release()


@pytest.mark.parametrize("name", ["cornell_16x12_r9", "cornell_teapot3_32x24_r4"])
def test_first_pass_initialises_the_accumulator(ctx, pkg, name):
    """mirt_render_first_pass = initAcu folded into the pass: over an accumulator full of junk it gives what zeroing + a normal pass
    gives (the r9 case defers a ninth of its samples to the exact kernel, which must start them from zero as well)."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture(name)
    fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"])
    fr.acu.write(np.random.default_rng(1).normal(size=sc.total_rays * 4).astype(np.float32))
    fr.execute_render(fresh=True)
    assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(fx["f_acu"]))
    assert np.array_equal(fr.seeds.read(np.int32), fx["f_seeds"])
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), fx["pixel"])
    fr.release()


def test_cornell_8M_samples_match_oracle(ctx, pkg):
    """The headline scene at 960x540, 16 rays per pixel, depth 8 (8.3 M samples -- about one deferred sample per 1.7 M goes to the
    exact kernel): fused pass == CPU oracle, every accumulator, seed and pixel."""
    from raytracing_amd.pyhost import render
    fx, sc0 = load_fixture("cornell_32x24_r4")
    sc = _variant(sc0, width=960, height=540, rays_per_pixel=16)
    seeds = A.make_seeds(sc.total_rays)
    orc = A.load_oracle()
    st = A.PassState(sc, seeds)
    A.run_pass(orc, sc, st, bounces=8)
    fr = render.FusedRenderer(ctx, sc, seeds=seeds)
    fr.execute_render(bounces=8)
    assert ctx.pass_deferred() > 0
    assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(st.acu))
    assert np.array_equal(fr.seeds.read(np.int32), st.seeds)
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), st.pixel)
    fr.release()


def test_headline_frame_two_exact_routes_agree(ctx, pkg):
    """BASELINE config 4 at its full size and depth (1920x1080, 256 rays per pixel, 8 bounces = 531 M samples): the default pair
    (optimistic kernel with every hand-written exact form -- 3-operation quotients, refined reciprocals, 9-operation sqrt, min/max box
    test -- plus the exact kernel for the deferred samples) against the single exact kernel, whose divisions and roots are the
    compiler's.  Per-pixel fp32 sums, RGBA8 and every ray's final seed must be identical."""
    import os
    from raytracing_amd.pyhost import render, scene
    from conftest import GOLDEN
    sc = scene.PackedScene(open(os.path.join(GOLDEN, "scene_cornell_1920x1080_r256.json")).read())
    out = []
    for exact_only in (False, True):
        ctx.set_exact_only(exact_only)
        try:
            fr = render.FusedRenderer(ctx, sc, want_radiance=True)
            fr.execute_render(bounces=8)
            deferred = ctx.pass_deferred()
            out.append((fr.radiance.read(np.float32), fr.pixel.read(np.uint8), fr.seeds.read(np.int32), deferred))
            fr.release()
        finally:
            ctx.set_exact_only(False)
    (r0, p0, s0, d0), (r1, p1, s1, d1) = out
    # a few hundred samples per frame leave the guard windows; the pass resolves its own pixels (later passes too, since round 4), so the exact kernel
    # re-runs the BLOCKS of 256 that hold one and the count is in whole blocks
    assert d1 == 0 and 0 < d0 < 1000 * 256 and d0 % 256 == 0
    assert np.array_equal(bits(r0), bits(r1)) and np.array_equal(p0, p1) and np.array_equal(s0, s1)
    assert (p0.reshape(-1, 4)[:, :3].max(axis=1) > 0).mean() > 0.9


def test_full_size_properties(ctx, pkg):
    """BASELINE config 4 geometry at 1920x1080 (rpp 4 to keep the test short): size-independent properties.
    (1) the 1080p frame's top-left 64x48 window... is NOT comparable (camera differs), so instead:
    (2) determinism: two runs give identical bytes; (3) tiling: 3 uneven row tiles == full frame;
    (4) w-channel counts: every ray's acu.w equals the number of shading events = integer in [0, 6*L+1]."""
    from raytracing_amd.pyhost import render, scene
    fx, sc0 = load_fixture("cornell_64x48_r1")
    sc = scene.PackedScene(sc0.d).resized(1920, 1080, 4)
    fr = render.FusedRenderer(ctx, sc)
    fr.execute_render()
    pix = fr.pixel.read(np.uint8)
    acu = fr.acu.read(np.float32).reshape(-1, 4)
    fr.release()
    fr2 = render.FusedRenderer(ctx, sc)
    fr2.execute_render()
    assert np.array_equal(fr2.pixel.read(np.uint8), pix)
    fr2.release()
    w = acu[:, 3]
    assert np.array_equal(w, np.round(w)) and w.min() >= 0 and w.max() <= 6 * len(sc.lights) + 1
    assert np.isfinite(acu).all() and (acu[:, :3] >= 0).all()
    parts = []
    for row0, nrows in ((0, 333), (333, 500), (833, 247)):
        t = render.FusedRenderer(ctx, sc, row0=row0, nrows=nrows, want_radiance=False)
        t.execute_render()
        parts.append(t.pixel.read(np.uint8))
        t.release()
    assert np.array_equal(np.concatenate(parts), pix)


def test_config5_tile_at_full_scale(ctx, pkg):
    """BASELINE config 5: 3840x2160, 1024 rays per pixel, thin lens, rows tiled over 8 GPUs.  One GPU's share here is cut down to
    a 24-row band (94 M rays, ids beyond 2^32 further down the frame): the band rendered alone must equal the same rows of a
    48-row tile, byte for byte (global ray ids, 64-bit index math), and a 3-row band at the very bottom of the frame -- ray ids
    around 8.5e9 -- must be deterministic."""
    import os
    from conftest import ROOT
    from raytracing_amd.pyhost import render, scene
    sc = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_3840x2160_r1024.json")).read())
    assert (sc.width, sc.height, sc.rpp) == (3840, 2160, 1024)
    big = render.FusedRenderer(ctx, sc, row0=1056, nrows=48, want_radiance=False)
    big.execute_render()
    pb = big.pixel.read(np.uint8).reshape(48, 3840, 4)
    big.release()
    small = render.FusedRenderer(ctx, sc, row0=1080, nrows=24, want_radiance=False)
    small.execute_render()
    ps = small.pixel.read(np.uint8).reshape(24, 3840, 4)
    small.release()
    assert np.array_equal(pb[24:], ps)
    assert ps[..., :3].max() > 0
    outs = []
    for _ in range(2):
        t = render.FusedRenderer(ctx, sc, row0=2157, nrows=3, want_radiance=False)
        t.execute_render()
        outs.append(t.pixel.read(np.uint8))
        t.release()
    assert np.array_equal(outs[0], outs[1]) and outs[0].reshape(-1, 4)[:, :3].max() > 0


def test_config5_one_gpus_share_at_full_size(ctx, pkg):
    """BASELINE config 5 as ONE of its eight GPUs sees it: the 270-row tile of 3840x2160 x 1024 rays per pixel, thin lens on --
    1 061 683 200 samples, 4.2 GB of seeds + 17 GB of accumulators resident -- rendered whole, through mirt_tile_rows' arithmetic.
    Tile 0 (rows 0..269) and tile 4 (rows 1080..1349, the image centre, ray ids beyond 2^32): each deterministic, a 48-row band cut
    out of it equal to that band rendered alone, integer sample counts in acu.w.  Times land in gpurun_out/config5_share.json.
    (The 32 x 32 lens grid itself is pinned against the reference binary in tests/test_ref_gpu.py.)"""
    import json
    import os
    from conftest import ROOT
    from raytracing_amd.pyhost import mirt, render, scene
    sc = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_3840x2160_r1024.json")).read())
    assert (sc.width, sc.height, sc.rpp) == (3840, 2160, 1024)
    ctx.set_profiling(True)
    report = {}
    for tile in (0, 4):
        r0, nr = C.c_uint32(), C.c_uint32()
        mirt.lib().mirt_tile_rows(sc.height, 8, tile, C.byref(r0), C.byref(nr))
        row0, nrows = r0.value, nr.value
        assert nrows == 270 and row0 == 270 * tile
        fr = render.FusedRenderer(ctx, sc, row0=row0, nrows=nrows, want_radiance=False)
        assert fr.nrays == 3840 * 270 * 1024
        fr.execute_render(bounces=5, fresh=True)
        ms5 = ctx.pass_timing()[0]
        pix = fr.pixel.read(np.uint8).reshape(nrows, sc.width, 4)
        for off in (0, 2 * fr.nrays, 4 * fr.nrays - (1 << 26)):            # three 256 MB windows of the 17 GB accumulator
            a = fr.acu.read(np.float32, count=1 << 26, offset=4 * off).reshape(-1, 4)
            assert np.array_equal(a[:, 3], np.round(a[:, 3])) and a[:, 3].min() >= 0 and a[:, 3].max() <= 7
            assert np.isfinite(a).all() and (a[:, :3] >= 0).all()
        ctx.seed_fill(fr.seeds, fr.first_ray, fr.nrays, 0)
        fr.passes = 1
        fr.execute_render(bounces=5, fresh=True)                            # again from the same seeds: the same frame
        assert np.array_equal(fr.pixel.read(np.uint8).reshape(nrows, sc.width, 4), pix)
        ctx.seed_fill(fr.seeds, fr.first_ray, fr.nrays, 0)
        fr.passes = 1
        fr.execute_render(bounces=8, fresh=True)
        ms8 = ctx.pass_timing()[0]
        fr.release()
        # without the 17 GB: the pass resolves its pixels block by block of a pixel's 1024 rays (four launches, FusedArgs::chunks) -- the same frame
        na = render.FusedRenderer(ctx, sc, row0=row0, nrows=nrows, want_radiance=False, keep_acu=False)
        na.execute_render(bounces=5, fresh=True)
        ms5_na = sum(ctx.pass_timing())
        assert np.array_equal(na.pixel.read(np.uint8).reshape(nrows, sc.width, 4), pix)
        na.release()
        ms5_sep = None
        if tile == 0:   # and the round-3 way, for the record: accumulators written, then read again by the separate copyToPixel
            os.environ["MIRT_INPASS_RESOLVE"] = "0"
            try:
                c2 = mirt.Context(0)
            finally:
                del os.environ["MIRT_INPASS_RESOLVE"]
            c2.set_profiling(True)
            f2 = render.FusedRenderer(c2, sc, row0=row0, nrows=nrows, want_radiance=False)
            f2.execute_render(bounces=5, fresh=True)
            ms5_sep = sum(c2.pass_timing())
            assert np.array_equal(f2.pixel.read(np.uint8).reshape(nrows, sc.width, 4), pix)
            f2.release()
            c2.destroy()
        band = render.FusedRenderer(ctx, sc, row0=row0 + 100, nrows=48, want_radiance=False)
        band.execute_render(bounces=5, fresh=True)
        assert np.array_equal(band.pixel.read(np.uint8).reshape(48, sc.width, 4), pix[100:148])
        band.release()
        assert pix[..., :3].max() > 0
        report[f"tile{tile}"] = {"row0": row0, "nrows": nrows, "samples": 3840 * 270 * 1024, "ms_5_bounces": round(ms5, 2), "ms_8_bounces": round(ms8, 2),
                                 "ms_5_bounces_no_acu_pass_plus_resolve": round(ms5_na, 2), "ms_5_bounces_separate_resolve_pass_plus_resolve": ms5_sep and round(ms5_sep, 2),
                                 "Msamples_s_5": round(3840 * 270 * 1024 / ms5 / 1e3, 1), "Msamples_s_8": round(3840 * 270 * 1024 / ms8 / 1e3, 1)}
    ctx.set_profiling(False)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, "config5_share.json"), "w"), indent=1)
    print("config5 share:", report)


def test_five_wave_variant_of_the_grid_kernels(pkg):
    """k_fusedPass<true, 1, 5> -- 96 registers, five waves per SIMD, nothing in scratch: the build launch_fused picks when a scene's cell tables
    leave room for five blocks per CU only -- forced for every grid scene (MIRT_GRID_WAVES=5, read once per process: a process of its own):
    frames, seeds and accumulators of the fixtures with grid meshes, tolerance 0."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MIRT_GRID_WAVES="5")
    env.pop("MIRT_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "grid_waves_check.py")], env=env, capture_output=True, text=True, timeout=600)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 10 and all(l["ok"] for l in lines), (lines, r.stderr[-1500:])


def test_error_paths(ctx, pkg):
    from raytracing_amd.pyhost import mirt
    with pytest.raises(mirt.MirtError) as e:
        ctx.kernel("noSuchKernel")
    assert e.value.code == -3
    k = ctx.kernel("initAcu")
    with pytest.raises(mirt.MirtError) as e:       # enqueue with unset args
        k.enqueue([64])
    assert e.value.code == -4
    small = ctx.buffer(64)
    k.set_args(small, np.array([1000], np.uint32))
    with pytest.raises(mirt.MirtError) as e:       # buffer too small for 1000 float4
        k.enqueue([1024])
    assert e.value.code == -5
    with pytest.raises(mirt.MirtError) as e:       # wrong scalar size
        k.set_arg(1, np.zeros(2, np.uint32))
    assert e.value.code == -1
    small.release()
    with pytest.raises(mirt.MirtError) as e:       # released buffer still bound
        k.enqueue([64])
    assert e.value.code == -2
    k.release()
    n, missing = ctx.program_check("__kernel void initAcu(__global float4* a){}\n// __kernel void ghost()\n__kernel void molTrace(){}")
    assert (n, missing) == (1, "molTrace")


def test_empty_row_tile_is_not_the_whole_frame(ctx, pkg):
    """more ranks than rows: a rank's tile has zero rows; its renderer owns nothing and its passes do nothing (it used to render the whole
    frame: `nrows or height`)."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("cornell_32x24_r4")
    t = render.FusedRenderer(ctx, sc, row0=sc.height, nrows=0)
    assert t.nrays == 0 and t.npix == 0
    t.execute_render()
    ctx.finish()
    t.release()
