import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")

FULL_CASES = [
    "basic_64x48_r1", "basic_32x24_r4", "triangles_64x48_r1", "triangles_32x24_r4",
    "cornell_64x48_r1", "cornell_32x24_r4", "cornell_16x12_r9",
    "cornell_teapot3_64x48_r1", "cornell_teapot3_32x24_r4", "cornell_official_64x48_r1",
    "twoLights_32x24_r4", "threeLights_32x24_r1",
    "basic2_32x24_r4", "cornell_teapot_32x24_r4", "cornell_teapot2_32x24_r4",   # with these, all ten of the reference's A10 scenes
    "own_studio_48x36_r4", "own_gems_48x36_r4", "own_gems_64x48_r1", "own_flat_32x24_r4",
]
OWN_SCENES = {"own_studio_48x36_r4": ("studio.xml", 48, 36, 4), "own_gems_48x36_r4": ("gems.xml", 48, 36, 4),
              "own_gems_64x48_r1": ("gems.xml", 64, 48, 1), "own_flat_32x24_r4": ("flat.xml", 32, 24, 4)}
PAGE = os.path.join(ROOT, "tests", "scenes", "page")
HOST = os.path.join(ROOT, "2015-raytracing_amd", "host")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def load_fixture(name):
    import a10_pass as A
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    sc = A.Scene(json.loads(bytes(fx["scene_json"]).decode()))
    return fx, sc


def bits(a):
    """bit patterns of a float32 array, every NaN mapped to one pattern: NaN payloads and signs are outside the numerics contract
    (x86, where the fixtures are made, and gfx950 propagate them differently; oracle/cl_numerics.h)"""
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32:
        return a
    u = a.view(np.uint32).copy()
    u[(u & 0x7FFFFFFF) > 0x7F800000] = 0x7FC00000
    return u


canon = bits


def assert_state_equal(tag, got, fx, prefix):
    """Compare a pass state (dict of numpy views: rays/shadow/pois structured, acu, seeds) with a fixture
    bit for bit.  Fields the reference leaves undefined are masked: o/d of dead rays (only mint/maxt are
    written, A10 code.cl:595, 647) and p/normal of vertices that were never hit."""
    assert np.array_equal(bits(got["acu"]).reshape(-1, 4), bits(fx[f"{prefix}_acu"])), f"{tag}: acu"
    assert np.array_equal(got["seeds"], fx[f"{prefix}_seeds"]), f"{tag}: seeds"
    for k in ("rays", "shadow"):
        for f in ("mint", "maxt"):
            assert np.array_equal(bits(got[k][f]), bits(fx[f"{prefix}_{k}_{f}"])), f"{tag}: {k}.{f}"
        # dead rays (mint == maxt == inf written by bouncePaths/initShadowTrace/lightRender) have undefined o/d
        live = ~(np.isinf(fx[f"{prefix}_{k}_mint"]) & np.isinf(fx[f"{prefix}_{k}_maxt"]))
        for f in ("o", "d"):
            assert np.array_equal(bits(got[k][f][live]), bits(fx[f"{prefix}_{k}_{f}"][live])), f"{tag}: {k}.{f}"
    assert np.array_equal(got["pois"]["matId"], fx[f"{prefix}_pois_matId"]), f"{tag}: matId"
    assert np.array_equal(bits(got["pois"]["atte"]), bits(fx[f"{prefix}_pois_atte"])), f"{tag}: atte"
    hit = fx[f"{prefix}_pois_matId"] >= 0
    for f in ("p", "normal"):
        assert np.array_equal(bits(got["pois"][f][hit]), bits(fx[f"{prefix}_pois_{f}"][hit])), f"{tag}: pois.{f}"


def assert_state_equals_pass_state(tag, got, st):
    """the same comparison against a live pass state (oracle/a10_pass.PassState: structured rays / shadow / pois, acu, seeds) instead of a fixture"""
    assert np.array_equal(bits(got["acu"]).reshape(-1, 4), bits(st.acu).reshape(-1, 4)), f"{tag}: acu"
    assert np.array_equal(got["seeds"], st.seeds), f"{tag}: seeds"
    for k, want in (("rays", st.rays), ("shadow", st.shadow)):
        for f in ("mint", "maxt"):
            assert np.array_equal(bits(got[k][f]), bits(want[f])), f"{tag}: {k}.{f}"
        live = ~(np.isinf(want["mint"]) & np.isinf(want["maxt"]))
        for f in ("o", "d"):
            assert np.array_equal(bits(got[k][f][live]), bits(want[f][live])), f"{tag}: {k}.{f}"
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"]), f"{tag}: matId"
    assert np.array_equal(bits(got["pois"]["atte"]), bits(st.pois["atte"])), f"{tag}: atte"
    hit = st.pois["matId"] >= 0
    for f in ("p", "normal"):
        assert np.array_equal(bits(got["pois"][f][hit]), bits(st.pois[f][hit])), f"{tag}: pois.{f}"
