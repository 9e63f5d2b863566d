"""The JavaScript host (2015-raytracing_amd/host/): scene ingest on CPU, rendering through the N-API
addon on the GPU.  Expected values come from the REFERENCE host code and kernels run in the build
container on the same scene files (oracle/gen/gen_golden.py, cases own_*): the fixtures' scene_json is
what the reference's loadScene/split*Data/Camera/Light produce for tests/scenes/page/."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import sys

from conftest import HOST, OWN_SCENES, PAGE, ROOT, bits, load_fixture

node = shutil.which("node")
pytestmark = pytest.mark.skipif(node is None, reason="node is not installed")
REF_PAGE = "/root/reference/Assign10-Path_Tracing"


def run_node(*args, stderr=False, **kw):
    r = subprocess.run([node] + list(args), capture_output=True, **kw)
    assert r.returncode == 0, r.stderr.decode()
    return r.stderr.decode() if stderr else r.stdout


def same_packed(a, b, path=""):
    if isinstance(a, dict):
        assert set(a) == set(b), f"{path}: keys {set(a) ^ set(b)}"
        for k in a:
            same_packed(a[k], b[k], path + "/" + k)
    elif isinstance(a, list) and a and isinstance(a[0], dict):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            same_packed(x, y, f"{path}[{i}]")
    elif isinstance(a, list):
        assert np.array_equal(np.asarray(a, np.float64), np.asarray(b, np.float64)), path
    else:
        assert a == b, path


@pytest.mark.parametrize("name", sorted(OWN_SCENES))
def test_pack_equals_reference_host(name):
    """camera / light packs, grid-sorted geometry, cell offsets, AABBs: exactly the reference host's arrays."""
    fx, _ = load_fixture(name)
    scene, w, h, rpp = OWN_SCENES[name]
    got = json.loads(run_node(os.path.join(HOST, "cli.js"), "pack", os.path.join(PAGE, "scenes", scene), str(w), str(h), str(rpp)))
    same_packed(json.loads(bytes(fx["scene_json"]).decode()), got)


@pytest.mark.gpu
def test_node_graph_replayed_passes(tmp_path):
    """`render --granular --graph`: five kernel-by-kernel passes through Node, the pass body replayed as one HIP graph from the third
    pass on (queue.captureBegin / captureEnd / launchGraph) == oracle."""
    import a10_pass as A
    fx, sc = load_fixture("own_gems_48x36_r4")
    seeds = A.make_seeds(sc.total_rays, seed_base=77)
    sfile = str(tmp_path / "seeds.i32")
    seeds.tofile(sfile)
    st = A.PassState(sc, seeds)
    orc = A.load_oracle()
    for p in range(5):
        A.run_pass(orc, sc, st, init_acu=(p == 0))
    out = str(tmp_path / "frame.rgba")
    run_node(os.path.join(HOST, "cli.js"), "render", os.path.join(PAGE, "scenes", "gems.xml"), "48", "36", "4", "5", out, "--granular", "--graph", "--seeds", sfile)
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(-1, 4), st.pixel)
    assert np.array_equal(bits(np.fromfile(out + ".radiance.f32", np.float32).reshape(-1, 4)), bits(A.radiance_sums(st.acu, 4)))


@pytest.mark.skipif(not os.path.isdir(REF_PAGE), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("scene", ["basic", "basic2", "cornell", "cornell_official", "cornell_teapot", "cornell_teapot2",
                                   "cornell_teapot3", "threeLights", "triangles", "twoLights"])
def test_reference_scenes_load_unchanged(scene):
    """scenes/*.xml and tri/*.json of the reference, read where they lie: our host and the reference's own host code
    (run in a vm sandbox) must emit identical buffers."""
    dump = os.path.join(os.path.dirname(HOST), "..", "oracle", "gen", "ref_host_dump.js")
    want = json.loads(run_node(dump, "/root/reference", scene + ".xml", "96", "64", "4", cwd="/tmp"))
    got = json.loads(run_node(os.path.join(HOST, "cli.js"), "pack", f"{REF_PAGE}/scenes/{scene}.xml", "96", "64", "4"))
    same_packed(want, got)


def test_grid_builder_drops_primitives_on_the_max_face():
    """min index clamped from below only, max index from above only (A10 code.js:948-953): a primitive lying entirely
    on the max face gets lo = n > hi = n-1 and lands in no cell -- the reference's Cornell_box_model quirk."""
    js = """
      const s = require(process.argv[1]);
      const b = new s.Bounds([0,0,0],[1,1,1]);
      const boxes = [[[0.1,0.1,0.1],[0.2,0.2,0.2]], [[1,0,0],[1,1,1]], [[0.4,0.4,0.4],[0.6,0.6,0.6]], [[-1,-1,-1],[2,2,2]]];
      const g = s.buildGrid(boxes.length, 2, b, (i) => boxes[i]);
      console.log(JSON.stringify({off: Array.from(g.offsets), order: Array.from(g.order)}));
    """
    out = json.loads(run_node("-e", js, os.path.join(HOST, "scene.js")))
    assert out["off"] == [0, 3, 5, 7, 9, 11, 13, 15, 17]
    assert 1 not in out["order"]                      # the primitive on x = 1 is dropped
    assert out["order"][:3] == [0, 2, 3]              # cell 0: input order, duplicates per cell
    assert out["order"].count(2) == 8 and out["order"].count(3) == 8 and out["order"].count(0) == 1


def test_xml_reader_handles_bom_comments_and_errors():
    js = """
      const s = require(process.argv[1]);
      const d = s.parseXML("\\ufeff<?xml version='1.0'?><a><!-- c <b>9</b> --><b> 1.5 </b><c><b>2</b></c></a>");
      const all = []; (function w(e){ for (const c of e.children) { if (c.name === 'b') all.push(Number(c.text)); w(c); } })(d);
      let err = ''; try { s.parseXML('<a><b></a>'); } catch (e) { err = e.message; }
      console.log(JSON.stringify({all: all, err: err}));
    """
    out = json.loads(run_node("-e", js, os.path.join(HOST, "scene.js")))
    assert out["all"] == [1.5, 2] and "unbalanced" in out["err"]


def test_webcl_surface_without_a_gpu():
    """The WebCL-shaped object exists, carries the constants the reference host uses, and refuses to run without an MI355X."""
    js = """
      const { webcl } = require(process.argv[1]);
      const need = ['DEVICE_NAME','DEVICE_TYPE','DEVICE_TYPE_ALL','DEVICE_TYPE_CPU','DEVICE_TYPE_GPU','KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE',
        'MEM_READ_ONLY','MEM_READ_WRITE','MEM_WRITE_ONLY','PLATFORM_NAME','PLATFORM_VENDOR','PLATFORM_VERSION','PLATFORM_PROFILE',
        'PLATFORM_EXTENSIONS','PROGRAM_BUILD_STATUS','PROGRAM_BUILD_LOG'];
      const missing = need.filter((k) => webcl[k] === undefined);
      const p = webcl.getPlatforms();
      const nd = p[0].getDevices(webcl.DEVICE_TYPE_ALL).length;
      let err = '';
      if (nd === 0) { try { webcl.createContext(); } catch (e) { err = e.name + ': ' + e.message; } }
      console.log(JSON.stringify({missing: missing, platforms: p.length, name: p[0].getInfo(webcl.PLATFORM_NAME), nd: nd, err: err}));
    """
    out = json.loads(run_node("-e", js, os.path.join(HOST, "webcl.js")))
    assert out["missing"] == [] and out["platforms"] == 1 and "MI355X" in out["name"]
    if out["nd"] == 0:
        assert "DEVICE_NOT_FOUND" in out["err"] and "no CPU fallback" in out["err"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fused", "fused+no-acu", "granular", "granular+fusion", "fused+device-grid", "granular+device-grid"])
@pytest.mark.parametrize("name", sorted(OWN_SCENES))
def test_node_render_matches_compiled_reference(tmp_path, name, mode):
    """scene.xml -> JS host -> N-API addon -> C ABI -> HIP kernels -> frame, against the compiled reference's frame.
    `device-grid`: the host only reads the files; mesh ingest (mirt_mesh_ingest), binning, mesh transforms and fp32 narrowing run on the
    device (mirt_grid_build / mirt_grid_gather_*)."""
    fx, sc = load_fixture(name)
    scene, w, h, rpp = OWN_SCENES[name]
    out = str(tmp_path / "frame.rgba")
    args = [os.path.join(HOST, "cli.js"), "render", os.path.join(PAGE, "scenes", scene), str(w), str(h), str(rpp), "1", out]
    if "granular" in mode:
        args.append("--granular")
    if "device-grid" in mode:
        args.append("--device-grid")
    if "no-acu" in mode:   # one pass, no per-ray accumulator: mirt_render_first_pass with acu == NULL resolves the frame inside the pass
        args.append("--no-acu")
    if "fusion" in mode:   # the same enqueues, recognised by the runtime and run as one fused launch per pass (mirt_ctx_set_fusion)
        args.append("--fusion")
    log = run_node(*args, stderr=True)
    pix = np.fromfile(out, np.uint8).reshape(-1, 4)
    rad = np.fromfile(out + ".radiance.f32", np.float32).reshape(-1, 4)
    assert np.array_equal(pix, fx["pixel"])
    assert np.array_equal(bits(rad), bits(fx["radiance"]))
    # the kernel-by-kernel modes really launch kernel by kernel (whoever made the context), `--fusion` really fuses
    fused = int(log.rsplit("fused from enqueues:", 1)[1].split()[0])
    assert fused == (1 if mode == "granular+fusion" else 0), log


def test_node_tile_arithmetic_matches_python():
    """mirt_tile_rows through the addon (no device needed) == pyhost/tiling.row_tiles: the Node host and the torch.distributed bench
    cut a frame the same way."""
    js = """
      const a = require(process.argv[1]);
      const out = [];
      for (const [h, n] of [[1080, 8], [1081, 8], [2160, 3], [7, 8], [1, 1]]) { const t = []; for (let i = 0; i < n; i++) { const r = a.tileRows(h, n, i); t.push([r.row0, r.nrows]); } out.push(t); }
      console.log(JSON.stringify(out));
    """
    import importlib.util
    spec = importlib.util.spec_from_file_location("tiling", os.path.join(os.path.dirname(HOST), "pyhost", "tiling.py"))
    tiling = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tiling)
    got = json.loads(run_node("-e", js, os.path.join(os.path.dirname(HOST), "mirt.node")))
    for (h, n), t in zip([(1080, 8), (1081, 8), (2160, 3), (7, 8), (1, 1)], got):
        want = tiling.row_tiles(h, n)
        assert [tuple(x) for x in t if x[1]] == [w for w in want if w[1]]
        assert sum(x[1] for x in t) == h


@pytest.mark.gpu
@pytest.mark.parametrize("mesh", ["icosphere.json", "octahedra.json", "terrain.json"])
def test_device_mesh_ingest_equals_host_parse(tmp_path, mesh):
    """parseMeshJSON (tri/meshDataVersion1.js:12-78) on the device (mirt_mesh_ingest: node x mesh transforms through the fp32 matrices in
    double, de-indexing, bounds over all vertices) == the JavaScript host's, whose packed scenes equal the reference host's: positions,
    normals (fp64 arrays of fp32-valued entries), bounds, material indices.  Meshes: one rotated + scaled + translated node, two nodes
    sharing a vertex array, an un-indexed mesh without nodes."""
    path = os.path.join(PAGE, "tri", mesh)
    h, d = str(tmp_path / "h"), str(tmp_path / "d")
    run_node(os.path.join(HOST, "cli.js"), "ingest", path, h)
    run_node(os.path.join(HOST, "cli.js"), "ingest", path, d, "--device")
    for ext in (".pos.f64", ".nor.f64"):
        a, b = np.fromfile(h + ext, np.uint64), np.fromfile(d + ext, np.uint64)
        assert a.size and np.array_equal(a, b), ext
    mh, md = json.load(open(h + ".meta.json")), json.load(open(d + ".meta.json"))
    assert mh == md


@pytest.mark.gpu
def test_device_mesh_ingest_rejects_a_bad_index(tmp_path):
    bad = {"meshes": [{"vertexPositions": [0, 0, 0, 1, 0, 0, 0, 1, 0], "vertexNormals": [0, 0, 1] * 3, "indices": [0, 1, 7], "materialIndex": 0}],
           "materials": [{"diffuseReflectance": [1, 1, 1, 1]}]}
    f = str(tmp_path / "bad.json")
    json.dump(bad, open(f, "w"))
    r = subprocess.run([node, os.path.join(HOST, "cli.js"), "ingest", f, str(tmp_path / "o"), "--device"], capture_output=True)
    assert r.returncode != 0 and b"refers past the 3 vertices" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [["--gpus", "1"], ["--gpus", "1", "--force-rccl"]])
def test_node_device_group_render(tmp_path, flags):
    """`render --gpus N`: webcl.createDeviceGroup -> one fused renderer per device on its row tile -> group.gather on device 0 ->
    one read-back.  On this one-GPU box N = 1 (tile == frame), once with the device copy and once through RCCL with a one-rank
    communicator (ncclSend / ncclRecv to self): the frame must equal the compiled reference's."""
    fx, sc = load_fixture("own_gems_48x36_r4")
    out = str(tmp_path / "frame.rgba")
    log = run_node(os.path.join(HOST, "cli.js"), "render", os.path.join(PAGE, "scenes", "gems.xml"), "48", "36", "4", "1", out, *flags, stderr=True)
    # the host reports how the gather moved the tile (mirt_gather_route through the addon): a device-local copy, or ncclSend/ncclRecv when forced
    assert ("gather routes per tile: rccl;" if "--force-rccl" in flags else "gather routes per tile: local;") in log, log
    assert "peer access root<-tile: 1" in log
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(-1, 4), fx["pixel"])
    assert np.array_equal(bits(np.fromfile(out + ".radiance.f32", np.float32).reshape(-1, 4)), bits(fx["radiance"]))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [["--gpus", "2"], ["--gpus", "8"], ["--gpus", "3", "--device-grid"], ["--gpus", "1", "--device-grid"]])
def test_node_device_group_render_n_tiles(tmp_path, flags):
    """`render --gpus N` with N > 1 on the one GPU of the box (MIRT_GROUP_ALLOW_REPEATED_DEVICES=1: the N contexts of the group share
    device 0): N fused renderers on N row tiles (36 rows over 8 = uneven), group.gather of the RGBA8 and radiance tiles on context 0,
    one read-back == the compiled reference's frame.  With --device-grid every tile's context ingests the mesh files and bins them on
    the device itself (buffers do not cross contexts; the loader's context is released when the tiles are built)."""
    fx, sc = load_fixture("own_gems_48x36_r4")
    out = str(tmp_path / "frame.rgba")
    log = run_node(os.path.join(HOST, "cli.js"), "render", os.path.join(PAGE, "scenes", "gems.xml"), "48", "36", "4", "1", out, *flags,
                   env=dict(os.environ, MIRT_GROUP_ALLOW_REPEATED_DEVICES="1"), stderr=True)
    n = int(flags[1])
    assert "gather routes per tile: " + " ".join(["local"] * n) + ";" in log, log   # every context shares device 0 in this rehearsal
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(-1, 4), fx["pixel"])
    assert np.array_equal(bits(np.fromfile(out + ".radiance.f32", np.float32).reshape(-1, 4)), bits(fx["radiance"]))


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(os.path.join(ROOT, "oracle", "_ref", "a10_gfx950_default.hsaco")) and os.path.exists(os.path.join(ROOT, "2015-raytracing_amd", "mirt_default.node"))),
                    reason="needs the reference's default build and mirt_default.node")
@pytest.mark.parametrize("flags", [[], ["--granular"], ["--gpus", "1"]])
def test_node_host_on_the_references_own_build_contract(tmp_path, flags):
    """MIRT_CONTRACT=default: host/webcl.js loads mirt_default.node -> libmirt_default.so, the kernels built as the reference's own host builds its program
    (program.build() without options, A10 code.js:599: 2.5-ulp division).  gems.xml (three grid meshes) through the Node CLI -- fused, kernel by kernel, and
    through a device group -- against the reference's code.cl compiled with ITS defaults and run on the GPU, same seeds: frame and radiance sums."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import a10_pass as A
    import ref_gpu as G
    fx, sc = load_fixture("own_gems_48x36_r4")
    seeds = np.asarray(fx["seeds_in"], np.int32)
    k = G.GpuRefKernels(os.path.join(ROOT, "oracle", "_ref", "a10_gfx950_default.hsaco"))
    st = A.PassState(sc, seeds)
    try:
        A.run_pass(k, sc, st)
    finally:
        k.release()
    sfile = str(tmp_path / "seeds.i32")
    seeds.tofile(sfile)
    out = str(tmp_path / "frame.rgba")
    run_node(os.path.join(HOST, "cli.js"), "render", os.path.join(PAGE, "scenes", "gems.xml"), "48", "36", "4", "1", out, "--seeds", sfile, *flags,
             env=dict(os.environ, MIRT_CONTRACT="default"))
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(-1, 4), st.pixel)
    assert np.array_equal(bits(np.fromfile(out + ".radiance.f32", np.float32).reshape(-1, 4)), bits(A.radiance_sums(st.acu, 4)))


@pytest.mark.gpu
def test_node_progressive_passes_and_explicit_seeds(tmp_path):
    """three passes with host-supplied seeds (the reference uploads a seed array, A10 code.js:1140-1154) == oracle."""
    import a10_pass as A
    fx, sc = load_fixture("own_studio_48x36_r4")
    seeds = A.make_seeds(sc.total_rays, seed_base=1234)
    sfile = str(tmp_path / "seeds.i32")
    seeds.tofile(sfile)
    st = A.PassState(sc, seeds)
    orc = A.load_oracle()
    for p in range(3):
        A.run_pass(orc, sc, st, init_acu=(p == 0))
    out = str(tmp_path / "frame.rgba")
    run_node(os.path.join(HOST, "cli.js"), "render", os.path.join(PAGE, "scenes", "studio.xml"), "48", "36", "4", "3", out, "--seeds", sfile)
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(-1, 4), st.pixel)
    assert np.array_equal(bits(np.fromfile(out + ".radiance.f32", np.float32).reshape(-1, 4)), bits(A.radiance_sums(st.acu, 4)))


@pytest.mark.skipif(not os.path.isdir(REF_PAGE), reason="reference tree not present (GPU box)")
def test_unmodified_reference_page_script_boots_on_our_webcl():
    """host/harness.js loads the reference's code.js + lib/ + tri/ UNMODIFIED with our `webcl`; without a GPU it must get as
    far as device discovery and scene loading (updateScene -> loadScene through our XHR/DOM stubs) and stop at 'no devices'."""
    js = """
      const h = require(process.argv[1]);
      const r = h.run(process.argv[2], 'cornell_teapot3.xml', 64, 48, 2, 1);
      const sb = h.makeSandbox(process.argv[2], {width: 64, height: 48});
      console.log(JSON.stringify({devices: r.devices, hasFrame: !!r.frame, fns: ['preRender','executeRender','loadScene','splitMeshData'].map((f) => typeof sb[f])}));
    """
    out = json.loads(run_node("-e", js, os.path.join(HOST, "harness.js"), REF_PAGE))
    assert out["fns"] == ["function"] * 4
    if out["devices"] == 0:
        assert out["hasFrame"] is False
