"""Randomly generated scenes through every path: the fused pass (optimistic pair and exact-only), the kernel-by-kernel path and the CPU oracle
must agree bit for bit.  The generator varies what the fixtures do not: how many sets there are, their grid resolutions (1..7 cells per axis,
loose sets included), how full the cells are (hundreds of slots in one cell, whole grids nearly empty), overlapping meshes, tiny and huge
primitives, one to three lights.  Geometry is binned by the restatement of the reference host's split*Data (tests/test_grid_build.py)."""
import os

import numpy as np
import pytest

import a10_pass as A
from conftest import bits, load_fixture
from test_grid_build import expected_grid
from test_gpu_parity import _variant, snapshot

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(pkg):
    from raytracing_amd.pyhost import mirt
    c = mirt.Context(0)
    yield c
    c.destroy()


def _bbox8(lo, hi):
    lo, hi = np.float32(lo), np.float32(hi)
    return [float(lo[0]), float(lo[1]), float(lo[2]), 1.0, float(hi[0]), float(hi[1]), float(hi[2]), 1.0]


def _tri_soup(rng, count, centre, spread, size):
    c = centre + rng.uniform(-spread, spread, size=(count, 1, 3))
    v = (c + rng.uniform(-size, size, size=(count, 3, 3))).astype(np.float32)
    n = rng.normal(size=(count, 3, 3))
    n = (n / np.linalg.norm(n, axis=2, keepdims=True)).astype(np.float32)
    return v, n


def _pack_tris(v, n, order):
    pos = np.zeros((len(order), 3, 4), np.float32)
    nor = np.zeros((len(order), 3, 4), np.float32)
    pos[:, :, :3] = v[order]
    nor[:, :, :3] = n[order]
    return pos.ravel().tolist(), nor.ravel().tolist()


def random_scene(base, seed):
    rng = np.random.default_rng(seed)
    d = dict(base.d)
    nmat = len(d["materials"]) // 4
    n_loose = int(rng.integers(1, 4))
    out = {"n_slabs": n_loose, "n_spheres": 0, "n_triangles": 0, "spheres": [], "s_matid": [], "s_box": [0] * (n_loose ** 3 + 1),
           "t_pos": [], "t_normal": [], "t_matid": [], "t_box": [0] * (n_loose ** 3 + 1)}
    ks = int(rng.integers(0, 7))
    if ks:
        c = rng.uniform(-0.8, 0.8, size=(ks, 3))
        r = rng.uniform(0.04, 0.35, size=(ks, 1))
        lo, hi = (c - r).min(axis=0) - 0.01, (c + r).max(axis=0) + 0.01
        b8 = _bbox8(lo, hi)
        off, order = expected_grid(0, np.concatenate([c, r], axis=1), [b8[0], b8[1], b8[2], b8[4], b8[5], b8[6]], n_loose)
        sph = np.concatenate([c, r * r], axis=1).astype(np.float32)
        out.update(n_spheres=len(order), spheres=sph[order].ravel().tolist(), s_matid=rng.integers(0, nmat, size=ks)[order].tolist(),
                   s_box=off.tolist(), sphere_bounds=b8)
    kt = int(rng.integers(0, 40))
    if kt:
        v, n = _tri_soup(rng, kt, np.zeros(3), 0.8, float(rng.choice([0.05, 0.3, 1.2])))
        b8 = _bbox8(v.reshape(-1, 3).min(axis=0) - 0.01, v.reshape(-1, 3).max(axis=0) + 0.01)
        off, order = expected_grid(1, v.reshape(kt, 9).astype(np.float64), [b8[0], b8[1], b8[2], b8[4], b8[5], b8[6]], n_loose)
        pos, nor = _pack_tris(v, n, order)
        out.update(n_triangles=len(order), t_pos=pos, t_normal=nor, t_matid=rng.integers(0, nmat, size=kt)[order].tolist(), t_box=off.tolist(),
                   triangle_bounds=b8)
    meshes = []
    for _ in range(int(rng.integers(0, 4))):
        T = int(rng.choice([1, 7, 40, 150, 400]))
        n_m = int(rng.integers(1, 8))
        v, n = _tri_soup(rng, T, rng.uniform(-0.5, 0.5, size=3), float(rng.choice([0.05, 0.3])), float(rng.choice([0.02, 0.1, 0.4])))
        b8 = _bbox8(v.reshape(-1, 3).min(axis=0), v.reshape(-1, 3).max(axis=0))
        off, order = expected_grid(1, v.reshape(T, 9).astype(np.float64), [b8[0], b8[1], b8[2], b8[4], b8[5], b8[6]], n_m)
        pos, nor = _pack_tris(v, n, order)
        meshes.append({"pos": pos, "normal": nor, "box": off.tolist(), "matid": int(rng.integers(0, nmat)), "bounds": b8, "nslabs": n_m,
                       "ntriangles": len(order)})
    out["meshes"] = meshes
    out["lights"] = [d["lights"][i % len(d["lights"])] for i in range(int(rng.integers(1, 4)))]
    w = int(os.environ.get("MIRT_SOAK_W", "48"))
    return _variant(base, width=w, height=max(1, w * 9 // 16), rays_per_pixel=int(rng.choice([1, 4])), **out)


# MIRT_SOAK=N: N scenes instead of 16 (a longer hunt for a rare disagreement; not part of the default run)
@pytest.mark.parametrize("seed", range(int(os.environ.get("MIRT_SOAK", "16"))))
def test_random_scene_all_paths_agree(ctx, pkg, seed):
    from raytracing_amd.pyhost import render
    _, base = load_fixture("cornell_teapot3_32x24_r4")
    sc = random_scene(base, 1000 + seed)
    seeds = A.make_seeds(sc.total_rays, seed_base=seed)
    st = A.PassState(sc, seeds)
    A.run_pass(A.load_oracle(), sc, st)
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
    assert np.array_equal(bits(got["acu"]), bits(st.acu)) and np.array_equal(got["seeds"], st.seeds), "kernel by kernel"
    assert np.array_equal(got["pois"]["matId"], st.pois["matId"])
    gr.release()
