#!/usr/bin/env python3
"""tests/default_contract_check.py -- run by tests/test_default_contract.py in a process of its own (MIRT_CONTRACT=default, so that pyhost loads
libmirt_default.so): the HIP path built for the reference's OWN build options -- program.build() without options (A10 code.js:599): AMD's default
2.5-ulp division and 3-ulp sqrt -- against the reference's code.cl compiled the same way (oracle/_ref/a10_gfx950_default.hsaco), device against
device, same seeds: every accumulator, seed and pixel.  Prints one JSON object per check; exits non-zero on the first difference, naming it."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import a10_pass as A  # noqa: E402
import ref_gpu as G  # noqa: E402
from conftest import canon, load_fixture  # noqa: E402

DEFAULT_HSACO = os.path.join(ROOT, "oracle", "_ref", "a10_gfx950_default.hsaco")
SCENES = ["basic_32x24_r4", "cornell_32x24_r4", "triangles_32x24_r4", "twoLights_32x24_r4", "threeLights_32x24_r1", "cornell_official_64x48_r1",
          "cornell_teapot3_32x24_r4", "own_flat_32x24_r4", "own_gems_48x36_r4", "own_studio_48x36_r4", "basic2_32x24_r4", "cornell_teapot_32x24_r4",
          "cornell_teapot2_32x24_r4"]


def first_difference(tag, got, want):
    g, w = canon(got).ravel(), canon(want).ravel()
    bad = np.flatnonzero(g != w)
    if bad.size == 0:
        return None
    i = int(bad[0])
    return f"{tag}: {bad.size} of {g.size} words differ; first at word {i}: ours 0x{int(g[i]):08x}, the reference's 0x{int(w[i]):08x}"


def scenes(mirt, render, scene):
    for name in SCENES:
        fx, sc0 = load_fixture(name)
        ps = scene.PackedScene(dict(sc0.d)).resized(480, 270, 16)
        sc = A.Scene(ps.d)
        seeds = A.make_seeds(sc.total_rays, seed_base=11)
        k = G.GpuRefKernels(DEFAULT_HSACO)
        st = A.PassState(sc, seeds)
        ctx = mirt.Context(0)
        try:
            for p in range(2):
                A.run_pass(k, sc, st, bounces=8, init_acu=(p == 0))
            for exact_only in (False, True):   # the optimistic pair (the default), then the exact kernel alone
                ctx.set_exact_only(exact_only)
                fr = render.FusedRenderer(ctx, ps, seeds=seeds)
                for p in range(2):
                    fr.execute_render(bounces=8)
                checks = (("accumulators", fr.acu.read(np.float32), st.acu), ("seeds", fr.seeds.read(np.int32), st.seeds), ("pixels", fr.pixel.read(np.uint8), st.pixel))
                fr.release()
                for tag, got, want in checks:
                    d = first_difference(f"{name} exact_only={exact_only} {tag}", np.asarray(got), np.asarray(want))
                    if d:
                        print(json.dumps({"check": "scenes", "scene": name, "ok": False, "difference": d}), flush=True)
                        return False
            print(json.dumps({"check": "scenes", "scene": name, "ok": True, "samples": 2 * sc.total_rays}), flush=True)
        finally:
            k.release()
            ctx.destroy()
    return True


def kernel_by_kernel(mirt, render, scene):
    """The kernel-by-kernel path of the same library (the reference's fourteen kernels one by one, as its host enqueues them; fusion off) against the
    reference's default build: every Ray, shadow Ray, Poi, accumulator and seed after one pass, on scenes with spheres, loose triangles and grid meshes."""
    from test_random_scenes import random_scene
    _, rbase = load_fixture("cornell_teapot3_32x24_r4")
    cases = [("basic_32x24_r4", None), ("cornell_teapot3_32x24_r4", None), ("own_gems_48x36_r4", None), ("twoLights_32x24_r4", None)]
    cases += [(f"generated scene {i}", i) for i in range(int(os.environ.get("MIRT_SOAK_GRANULAR", "24")))]
    for name, gen in cases:
        if gen is None:
            fx, sc0 = load_fixture(name)
            ps = scene.PackedScene(dict(sc0.d)).resized(96, 54, 4)
            sc = A.Scene(ps.d)
        else:   # (sphere and triangle sets in grids of 1..3 cells per axis, meshes in 1..7: tests/test_random_scenes.py)
            sc = random_scene(rbase, 1000 + gen, rpp=4)
            ps = sc
        seeds = A.make_seeds(sc.total_rays, seed_base=5)
        k = G.GpuRefKernels(DEFAULT_HSACO)
        st = A.PassState(sc, seeds)
        A.run_pass(k, sc, st)
        ctx = mirt.Context(0)
        ctx.set_fusion(0)
        gr = render.GranularRenderer(ctx, ps, seeds=seeds)
        try:
            gr.execute_render()
            rays, shadow, pois = gr.read("rays").view(A.RAY_DT), gr.read("shadow").view(A.RAY_DT), gr.read("pois").view(A.POI_DT)
            checks = [("accumulators", gr.read("acu"), st.acu), ("seeds", gr.read("seeds"), st.seeds), ("pixels", gr.read("pixel"), st.pixel),
                      ("matId", pois["matId"], st.pois["matId"]), ("atte", pois["atte"], st.pois["atte"])]
            hit = st.pois["matId"] >= 0   # (p / normal of a vertex never hit, o / d of a dead ray: fields the reference leaves undefined, conftest.assert_state_equal)
            checks += [("p", pois["p"][hit], st.pois["p"][hit]), ("normal", pois["normal"][hit], st.pois["normal"][hit])]
            for tag, g_, w_ in (("rays", rays, st.rays), ("shadow rays", shadow, st.shadow)):
                live = ~(np.isinf(w_["mint"]) & np.isinf(w_["maxt"]))
                checks += [(tag + " mint", g_["mint"], w_["mint"]), (tag + " maxt", g_["maxt"], w_["maxt"]), (tag + " o", g_["o"][live], w_["o"][live]), (tag + " d", g_["d"][live], w_["d"][live])]
            for tag, got, want in checks:
                d = first_difference(f"{name} kernel by kernel, {tag}", np.ascontiguousarray(got), np.ascontiguousarray(want))
                if d:
                    print(json.dumps({"check": "granular", "scene": name, "ok": False, "difference": d}), flush=True)
                    return False
            print(json.dumps({"check": "granular", "scene": name, "ok": True}), flush=True)
        finally:
            k.release()
            gr.release()
            ctx.destroy()
    return True


def frames(mirt, render, scene):
    """BASELINE configs 1-3 and the molecule mode at full size: the Assign01 / 04 / 07 frame kernels of the default-contract library against the reference's
    code.cl built with ITS defaults (oracle/_ref/a0N_gfx950_default.hsaco), on the device: every pixel, and every ray's maxt."""
    import frame_pass as F
    from test_frames import fixture, resized
    ctx = mirt.Context(0)
    try:
        for name, size in (("frame_a01_512x512", None), ("frame_a04_parliament_96x64", (1024, 1024)), ("frame_a04_teapot_160x120", (1024, 1024)),
                           ("frame_a07_parliament_n16_160x120", (1920, 1080)), ("frame_a07_teapot_n2_160x120", (1920, 1080)), ("frame_a07_teapot_n8_160x120", (1920, 1080)),
                           ("frame_a07_mol_3IZ4_n16_96x64", (1920, 1080)), ("frame_a07_mol_c60_n4_160x120", (1920, 1080)), ("frame_a07_own_terrain_n5_96x64", (1920, 1080)),
                           ("frame_a04_house_160x120", (1024, 1024)), ("frame_a07_house_n2_160x120", (1920, 1080)), ("frame_a07_house_n8_160x120", (1920, 1080))):
            _, d = fixture(name)
            if size:
                d = resized(d, *size)
            want_px, want_rays = F.run_frame_gpu(F.Frame(d), default_build=True)
            px, rays = render.render_frame(ctx, render.FramePacked(d))
            diff = first_difference(name + " pixels", np.ascontiguousarray(px), np.ascontiguousarray(want_px))
            if not diff and want_rays is not None and rays is not None:
                got = np.ascontiguousarray(rays).view(A.RAY_DT)
                diff = first_difference(name + " rays.maxt", np.ascontiguousarray(got["maxt"]), np.ascontiguousarray(want_rays["maxt"]))
            if diff:
                print(json.dumps({"check": "frames", "frame": name, "ok": False, "difference": diff}), flush=True)
                return False
            print(json.dumps({"check": "frames", "frame": name, "ok": True}), flush=True)
    finally:
        ctx.destroy()
    return True


def big_pixels(mirt, render, scene):
    """1024 rays per pixel (BASELINE config 5's 32 x 32 lens grid): the pass resolving its pixels block by block, four launches, without an accumulator"""
    ctx = mirt.Context(0)
    k = G.GpuRefKernels(DEFAULT_HSACO)
    try:
        for name in ("cornell_32x24_r4", "cornell_teapot3_32x24_r4"):
            fx, sc0 = load_fixture(name)
            ps = scene.PackedScene(dict(sc0.d)).resized(6, 4, 1024)
            sc = A.Scene(ps.d)
            seeds = A.make_seeds(sc.total_rays, seed_base=3)
            st = A.PassState(sc, seeds)
            A.run_pass(k, sc, st, bounces=8)
            fr = render.FusedRenderer(ctx, ps, seeds=seeds, keep_acu=False)
            fr.execute_render(bounces=8, fresh=True)
            checks = (("seeds", fr.seeds.read(np.int32), st.seeds), ("pixels", fr.pixel.read(np.uint8), st.pixel),
                      ("radiance", fr.radiance.read(np.float32), A.radiance_sums(st.acu, 1024)))
            fr.release()
            for tag, got, want in checks:
                d = first_difference(f"{name} x 1024 rays per pixel, {tag}", np.asarray(got), np.asarray(want))
                if d:
                    print(json.dumps({"check": "big_pixels", "scene": name, "ok": False, "difference": d}), flush=True)
                    return False
            print(json.dumps({"check": "big_pixels", "scene": name, "ok": True}), flush=True)
    finally:
        k.release()
        ctx.destroy()
    return True


def random_scenes(mirt, render, scene, count):
    """`count` generated scenes (tests/test_random_scenes.py: one to three loose and grid sets, grids of 1..7 cells per axis, crowded and empty cells, one to
    three lights) -- the optimistic pair, the exact kernel alone and the pass that resolves its own pixels, each against the reference's default build."""
    from test_random_scenes import random_scene
    _, base = load_fixture("cornell_teapot3_32x24_r4")
    k = G.GpuRefKernels(DEFAULT_HSACO)
    ctx = mirt.Context(0)
    try:
        for seed in range(count):
            sc = random_scene(base, 1000 + seed, rpp=4)   # (four rays per pixel: at one, the reference's initTrace races on seeds[column] on a GPU)
            seeds = A.make_seeds(sc.total_rays, seed_base=seed)
            st = A.PassState(sc, seeds)
            A.run_pass(k, sc, st)
            for exact_only in (False, True):
                ctx.set_exact_only(exact_only)
                fr = render.FusedRenderer(ctx, sc, seeds=seeds)
                fr.execute_render()
                checks = [("accumulators", fr.acu.read(np.float32), st.acu), ("seeds", fr.seeds.read(np.int32), st.seeds), ("pixels", fr.pixel.read(np.uint8), st.pixel)]
                fr.release()
                fr = render.FusedRenderer(ctx, sc, seeds=seeds, keep_acu=False)
                fr.execute_render(fresh=True)
                checks += [("seeds (in-pass resolve)", fr.seeds.read(np.int32), st.seeds), ("pixels (in-pass resolve)", fr.pixel.read(np.uint8), st.pixel)]
                fr.release()
                for tag, got, want in checks:
                    d = first_difference(f"random scene {seed}, exact_only={exact_only}, {tag}", np.asarray(got), np.asarray(want))
                    if d:
                        print(json.dumps({"check": "random", "seed": seed, "ok": False, "difference": d}), flush=True)
                        return False
            if seed % 100 == 99:
                print(json.dumps({"check": "random", "ok": True, "scenes_so_far": seed + 1}), flush=True)
        print(json.dumps({"check": "random", "ok": True, "scenes": count}), flush=True)
    finally:
        ctx.set_exact_only(False)
        k.release()
        ctx.destroy()
    return True


def headline(mirt, render, scene):
    """BASELINE's headline frame at full size: 1920x1080 x 256 rays per pixel, depth 8 (530 841 600 samples): the optimistic pair, then the exact kernel alone."""
    import ctypes as C
    base = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
    sc = A.Scene(base.d)
    n, npix = sc.total_rays, sc.width * sc.height
    ctx = mirt.Context(0)
    fr = render.FusedRenderer(ctx, base, want_radiance=False)
    seeds = fr.seeds.read(np.int32)
    fr.release()
    k = G.GpuRefKernels(DEFAULT_HSACO)

    class St:
        pass
    st = St()
    st.rays, st.pois, st.shadow = G.DevBuf(n * 48), G.DevBuf(n * 64), G.DevBuf(n * 48)
    st.acu, st.seeds, st.pixel = G.DevBuf(n * 16), G.DevBuf(n * 4), G.DevBuf(npix * 4)
    st.passes = 1
    st.seeds.upload(seeds)
    del seeds
    A.run_pass(k, sc, st, bounces=8)
    for b in (st.rays, st.pois, st.shadow):
        b.free()
    want_pixel, want_seeds = st.pixel.download(np.uint8, npix * 4), st.seeds.download(np.int32, n)
    try:
        for exact_only in (False, True):
            ctx.set_exact_only(exact_only)
            fr = render.FusedRenderer(ctx, base, want_radiance=False)   # (the same seed fill as the one the reference started from)
            fr.execute_render(bounces=8, fresh=True)
            ctx.finish()
            try:
                for tag, got, want in (("pixels", fr.pixel.read(np.uint8), want_pixel), ("seeds", fr.seeds.read(np.int32), want_seeds)):
                    d = first_difference(f"headline exact_only={exact_only} {tag}", got, want)
                    if d:
                        print(json.dumps({"check": "headline", "ok": False, "difference": d}), flush=True)
                        return False
                chunk = 1 << 26
                for off in range(0, 4 * n, chunk):
                    m = min(chunk, 4 * n - off)
                    want = np.empty(m, np.float32)
                    G.chk(G.hip().hipMemcpy(want.ctypes.data_as(C.c_void_p), C.c_void_p(st.acu.ptr + 4 * off), 4 * m, 2), "D2H")
                    d = first_difference(f"headline exact_only={exact_only} accumulators [{off}, {off + m})", fr.acu.read(np.float32, count=m, offset=4 * off), want)
                    if d:
                        print(json.dumps({"check": "headline", "ok": False, "difference": d}), flush=True)
                        return False
            finally:
                fr.release()
    finally:
        for b in (st.acu, st.seeds, st.pixel):
            b.free()
        k.release()
        ctx.set_exact_only(False)
        ctx.destroy()
    print(json.dumps({"check": "headline", "ok": True, "samples": n}), flush=True)
    return True


def main():
    assert os.environ.get("MIRT_CONTRACT") == "default", "run with MIRT_CONTRACT=default"
    graft.load_package()
    from raytracing_amd.pyhost import mirt, render, scene
    assert mirt.LIB_PATH.endswith("libmirt_default.so"), mirt.LIB_PATH
    which = sys.argv[1:] or ["scenes", "headline"]
    ok = True
    if "scenes" in which:
        ok = scenes(mirt, render, scene) and ok
    if ok and "big_pixels" in which:
        ok = big_pixels(mirt, render, scene) and ok
    if ok and "frames" in which:
        ok = frames(mirt, render, scene) and ok
    if ok and "granular" in which:
        ok = kernel_by_kernel(mirt, render, scene) and ok
    if ok and "random" in which:
        ok = random_scenes(mirt, render, scene, int(os.environ.get("MIRT_SOAK", "64"))) and ok
    if ok and "headline" in which:
        ok = headline(mirt, render, scene) and ok
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
