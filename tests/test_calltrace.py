"""The reference host's OWN call stream on this runtime.

tests/golden/calltrace_*.json (+ .bin.gz) are recordings of the reference's UNMODIFIED Assign10 code.js
(preRender -> executeRender x passes -> postRender, A10 code.js:1784-1859; real getLocalWS shapes :645-672, setArg
typed-array forms :1124-1131, non-blocking writes :1153, 1183-1185, release order :1539-1552) made in the build
container on the recording WebCL (host/webcl_record.js, oracle/gen/record_calltrace.js, Math.random pinned).
calltrace_*_expect.npz is what that stream reads back when its kernels are the reference's own code.cl compiled for
x86 (oracle/_ref; oracle/gen/gen_calltrace.py).  Numbers and kernel names only.

  CPU:  the committed traces are what the reference issues (re-recorded where the reference tree exists), the replayer
        re-issues a trace unchanged, our CPU restatement reproduces the expectations from the same stream, and
        host/renderer.js (our re-write of the sequence) issues the same launches with the same scalar arguments.
  GPU:  node host/replay.js replays each trace through webcl.js -> mirt.node -> libmirt.so -> HIP kernels and every
        read-back, the accumulator and the seed buffer must equal the expectations bit for bit.
"""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import calltrace as CT
from conftest import GOLDEN, HOST, ROOT, bits

node = shutil.which("node")
pytestmark = pytest.mark.skipif(node is None, reason="node is not installed")
REFROOT = "/root/reference"
CASES = {  # name -> (scene, width, height, sqrt rays per pixel, passes, Math.random seed)
    "cornell_320x240_k1_p2": ("cornell.xml", 320, 240, 1, 2, 20150410),
    "cornell_320x240_k2_p2": ("cornell.xml", 320, 240, 2, 2, 20150411),
    "cornell_teapot3_320x240_k1_p2": ("cornell_teapot3.xml", 320, 240, 1, 2, 20150412),
    "threeLights_160x120_k3_p3": ("threeLights.xml", 160, 120, 3, 3, 20150413),
}


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def run_node(*args, **kw):
    r = subprocess.run([node] + list(args), capture_output=True, **kw)
    assert r.returncode == 0, r.stderr.decode()
    return r.stdout


def check_against_expectation(name, reads, acu, seeds):
    ex = np.load(CT.golden_prefix(name) + "_expect.npz")
    k = CASES[name][3]
    sizes = ex["read_sizes"].tolist()
    assert [len(r) for r in reads] == sizes, "read-back sizes"
    want = ex["reads"].tobytes()
    off = 0
    for i, (r, n) in enumerate(zip(reads, sizes)):
        w = want[off:off + n]
        off += n
        if r != w:
            a, b = np.frombuffer(r, np.uint8), np.frombuffer(w, np.uint8)
            bad = np.flatnonzero(a != b)
            raise AssertionError(f"{name}: read-back {i} ({n} bytes) differs in {bad.size} bytes, first at {bad[0]}: {a[bad[0]]} != {b[bad[0]]}")
    import a10_pass as A
    acu = np.frombuffer(acu, np.float32).reshape(-1, 4)
    rad = A.radiance_sums(acu, k * k)
    assert np.array_equal(bits(rad), bits(ex["radiance"])), f"{name}: per-pixel radiance sums"
    assert np.array_equal(sha(acu), ex["sha_acu"]), f"{name}: accumulator"
    assert np.array_equal(sha(np.frombuffer(seeds, np.uint8)), ex["sha_seeds"]), f"{name}: seeds"


# ---------------------------------------------------------------- CPU ----------------------------------------------------------------

@pytest.mark.skipif(not os.path.isdir(REFROOT), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", sorted(CASES))
def test_committed_trace_is_what_the_unmodified_page_script_issues(name, tmp_path):
    scene, w, h, k, passes, rseed = CASES[name]
    out = str(tmp_path / "t")
    run_node(os.path.join(ROOT, "oracle", "gen", "record_calltrace.js"), REFROOT, scene, str(w), str(h), str(k), str(passes), str(rseed), out, cwd="/tmp")
    t, blob = CT.load_trace(out)
    t0, blob0 = CT.load_trace(CT.golden_prefix(name))
    assert t["events"] == t0["events"] and blob == blob0 and t["meta"] == t0["meta"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_trace_shapes_are_the_reference_hosts(name):
    """What the stream looks like: the structure SURVEY.md section 3 derives from code.js, asserted on the recorded data."""
    t, _ = CT.load_trace(CT.golden_prefix(name))
    scene, w, h, k, passes, _ = CASES[name]
    n = w * h * k * k
    ev = t["events"]
    shapes = CT.launch_shapes(t)
    # 2-D initTrace with local [8,8] (getLocalWS(2) on a multiple of 64) and globals padded to it; everything else 1-D [64], padded
    it = [s for s in shapes if s[0] == "initTrace"]
    assert len(it) == passes and all(s == ("initTrace", 2, (-(-w // 8) * 8, -(-h // 8) * 8), (8, 8)) for s in it)
    for nm, dim, g, l in shapes:
        if nm in ("sizeofRay", "sizeofPoi"):
            assert (dim, g, l) == (1, (1,), (1,))
        elif nm == "copyToPixel":
            assert (dim, g, l) == (1, (-(-(w * h) // 64) * 64,), (64,))
        elif nm != "initTrace":
            assert (dim, g, l) == (1, (-(-n // 64) * 64,), (64,)), nm
    # every write is non-blocking with an empty wait list and directly precedes further enqueues without a finish in between
    writes = [e for e in ev if e["op"] == "enqueueWriteBuffer"]
    assert writes and all(e["blocking"] is False and e["waitList"] == 0 and e["offset"] == 0 for e in writes)
    # scalar setArg forms: 1-element Uint32Array / Float32Array, float16 = 16 floats, AABB = 8 floats (code.js:610-621, 1124-1131)
    forms = {(e["type"], e["bytes"]) for e in ev if e["op"] == "setArg" and "buffer" not in e}
    assert forms == {("Uint32Array", 4), ("Float32Array", 4), ("Float32Array", 32), ("Float32Array", 64)}
    # three struct-size probes (Ray, Poi, Ray), one frame read-back per pass into the canvas' Uint8ClampedArray
    reads = [e for e in ev if e["op"] == "enqueueReadBuffer"]
    assert [e.get("answer") for e in reads[:3]] == [48, 64, 48] and len(reads) == 3 + passes
    assert all(e["type"] == "Uint8ClampedArray" and e["bytes"] == w * h * 4 and e["blocking"] is False for e in reads[3:])
    # releaseCLResources pops its stack: the context is created first and released last (code.js:1539-1552)
    rel = [e for e in ev if e["op"] == "release"]
    assert rel[-1]["kind"] == "context" and rel[-2]["kind"] == "queue"
    created = {e["id"] for e in ev if e["op"] in ("createBuffer", "createKernel", "createProgram", "createCommandQueue", "createContext")}
    # ... and everything else exactly once, except the bouncePaths kernel: prepareBouncePaths never pushes it on the release stack
    # (code.js:1440-1455), so the context goes away with one kernel still alive -- the runtime must survive that (mirt_ctx_destroy
    # reaps the children of a context)
    leaked = created - {e["id"] for e in rel}
    assert len(rel) == len(created) - 1 and len(leaked) == 1
    assert [e["name"] for e in ev if e["op"] == "createKernel" and e["id"] in leaked] == ["bouncePaths"]
    # finish(): after initAcu, after each size probe, after every sceneRender, after each frame read
    n_lights = sum(1 for s in shapes if s[0] == "lightRender") // passes
    assert sum(1 for e in ev if e["op"] == "finish") == 1 + 3 + passes * (6 * n_lights + 1)


@pytest.mark.parametrize("name", sorted(CASES))
def test_replayer_reissues_the_stream_unchanged(name):
    """host/replay.js pointed at a second recorder: the re-recorded stream must equal the trace, event for event."""
    js = """
      const { replay, load } = require(process.argv[1]);
      const { makeRecordingWebCL } = require(process.argv[2]);
      const { trace, blob } = load(process.argv[3]);
      const rec = makeRecordingWebCL(trace.meta);
      const r = replay(trace, blob, { webcl: rec.webcl, noDumps: true });
      const t2 = rec.trace();
      // the recorder fingerprints the program text; the replayer hands over the manifest form of the same kernel list
      const strip = (e) => { const c = Object.assign({}, e); delete c.sourceBytes; delete c.sourceSha256; return c; };
      const same = JSON.stringify(trace.events.map(strip)) === JSON.stringify(t2.events.map(strip)) && Buffer.compare(blob, rec.blob()) === 0;
      console.log(JSON.stringify({ same: same, leaked: r.leaked, checked: r.checked }));
    """
    out = json.loads(run_node("-e", js, os.path.join(HOST, "replay.js"), os.path.join(HOST, "webcl_record.js"), CT.golden_prefix(name)))
    assert out["same"] and out["leaked"] == 1 and out["checked"]["structSizes"] == 3 and out["checked"]["preferredMultiple"] > 0


@pytest.mark.parametrize("name", sorted(CASES))
def test_restatement_reproduces_the_expectations_from_the_same_stream(name):
    """oracle/liboracle.so (our C restatement) driven by the reference host's stream == oracle/_ref driven by it."""
    import a10_pass as A
    t, blob = CT.load_trace(CT.golden_prefix(name))
    res = CT.play(t, blob, A.load_oracle())
    check_against_expectation(name, res["reads"], res["buffers"]["acu"].tobytes(), res["buffers"]["seeds"].tobytes())


@pytest.mark.skipif(not os.path.isdir(REFROOT), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", ["cornell_320x240_k2_p2", "cornell_teapot3_320x240_k1_p2", "threeLights_160x120_k3_p3"])
def test_our_renderer_issues_the_same_launches_as_the_reference_host(name):
    """host/renderer.js::GranularRenderer is our re-write of the sequence.  On the recorder, for the same scene file, it must issue the
    same launches in the same order as the reference's code.js did: kernel, dim, global, local, and for every launch the same scalar
    argument bytes and the same buffer sizes bound at launch time (buffer CONTENTS are compared by test_js_host.py)."""
    scene, w, h, k, passes, _ = CASES[name]
    js = """
      const path = require('path');
      const host = process.argv[1];
      const { makeRecordingWebCL } = require(path.join(host, 'webcl_record.js'));
      const R = require(path.join(host, 'renderer.js')), S = require(path.join(host, 'scene.js'));
      const [file, w, h, k, passes] = [process.argv[2], +process.argv[3], +process.argv[4], +process.argv[5], +process.argv[6]];
      const rec = makeRecordingWebCL({});
      R.setWebCL(rec.webcl);
      const packed = S.packScene(S.loadSceneFile(file, w, h), w, h, k * k);
      const g = new R.GranularRenderer(packed, { seeds: new Int32Array(w * h * k * k).fill(1) });
      for (let i = 0; i < passes; i++) { g.executeRender(5); g.readPixels(); }
      g.release();
      console.log(JSON.stringify(rec.trace()));
    """
    ours = json.loads(run_node("-e", js, HOST, f"{REFROOT}/Assign10-Path_Tracing/scenes/{scene}", str(w), str(h), str(k), str(passes)))
    theirs, _ = CT.load_trace(CT.golden_prefix(name))

    def launches(t):
        size, kern, out = {}, {}, []
        for e in t["events"]:
            if e["op"] == "createBuffer":
                size[e["id"]] = e["bytes"]
            elif e["op"] == "createKernel":
                kern[e["id"]] = {}
            elif e["op"] == "setArg":
                kern[e["kernel"]][e["index"]] = ("buf", max(size[e["buffer"]], 16)) if "buffer" in e else ("val", e["hex"])
            elif e["op"] == "enqueueNDRangeKernel":
                out.append((e["name"], e["dim"], e["global"], e["local"], sorted(kern[e["kernel"]].items())))
        return out

    a, b = launches(ours), launches(theirs)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, f"launch {i}: ours {x} != reference host's {y}"


# ---------------------------------------------------------------- GPU ----------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_replayed_reference_host_stream_matches_compiled_reference(name, tmp_path):
    """The reference host's recorded call stream, replayed call for call through webcl.js -> mirt.node -> libmirt.so -> HIP:
    every frame it reads back, its accumulator and its seed buffer == the compiled reference kernels run on the same stream.
    MIRT_FUSION=0: every enqueue launches (webcl.createContext fuses whole passes by default: tests/test_fusion.py runs that)."""
    out = str(tmp_path / "r")
    rep = json.loads(run_node(os.path.join(HOST, "replay.js"), CT.golden_prefix(name), out, env=dict(os.environ, MIRT_FUSION="0")))
    assert rep["fusedPasses"] == 0, rep
    assert rep["leaked"] == 1 and rep["checked"]["structSizes"] == 3, rep   # the page script never releases its bouncePaths kernel
    ex = np.load(CT.golden_prefix(name) + "_expect.npz")
    raw = open(out + ".reads.bin", "rb").read()
    reads, off = [], 0
    for n in ex["read_sizes"].tolist():
        reads.append(raw[off:off + n])
        off += n
    assert off == len(raw)
    check_against_expectation(name, reads, open(out + ".acu.bin", "rb").read(), open(out + ".seeds.bin", "rb").read())


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(os.path.join(ROOT, "oracle", "_ref", "a10_gfx950_default.hsaco")) and os.path.exists(os.path.join(ROOT, "2015-raytracing_amd", "mirt_default.node"))),
                    reason="needs the reference's default build and mirt_default.node")
@pytest.mark.parametrize("fusion", ["0", "2"])
@pytest.mark.parametrize("name", ["cornell_320x240_k2_p2", "threeLights_160x120_k3_p3"])
def test_replayed_stream_on_the_references_own_build_contract(name, fusion, tmp_path):
    """The same recorded streams of the unmodified page script under the SECOND numerics contract: replayed through webcl.js -> mirt_default.node ->
    libmirt_default.so (MIRT_CONTRACT=default), kernel by kernel and fused by the runtime, against the stream played on the reference's code.cl as its own
    host builds it (program.build() without options; oracle/_ref/a10_gfx950_default.hsaco, on the GPU): every read-back, the accumulator, the seeds.
    (The two traces with more than one ray per pixel: at one, the reference's initTrace races on seeds[column] on a GPU, oracle/ref_gpu.py.)"""
    import ref_gpu as G
    trace, blob = CT.load_trace(CT.golden_prefix(name))
    k = G.GpuRefKernels(os.path.join(ROOT, "oracle", "_ref", "a10_gfx950_default.hsaco"))
    try:
        want = CT.play(trace, blob, k)
    finally:
        k.release()
    out = str(tmp_path / "r")
    rep = json.loads(run_node(os.path.join(HOST, "replay.js"), CT.golden_prefix(name), out, env=dict(os.environ, MIRT_FUSION=fusion, MIRT_CONTRACT="default")))
    assert (rep["fusedPasses"] > 0) == (fusion == "2"), rep
    raw = open(out + ".reads.bin", "rb").read()
    assert raw == b"".join(want["reads"]), "read-backs"
    assert open(out + ".acu.bin", "rb").read() == want["buffers"]["acu"].tobytes(), "accumulator"
    assert open(out + ".seeds.bin", "rb").read() == want["buffers"]["seeds"].tobytes(), "seeds"
