import os, sys, json, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"oracle")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import __graft_entry__ as graft
import a10_pass as A, ref_gpu as G
from conftest import canon, load_fixture
graft.load_package()
from raytracing_amd.pyhost import mirt, render, scene
from test_random_scenes import random_scene
_, base = load_fixture("cornell_teapot3_32x24_r4")
k = G.GpuRefKernels(os.path.join(ROOT,"oracle","_ref","a10_gfx950_default.hsaco"))
ctx = mirt.Context(0)
for seed in range(int(sys.argv[1]) if len(sys.argv)>1 else 12):
    sc = random_scene(base, 1000+seed, rpp=4)
    seeds = A.make_seeds(sc.total_rays, seed_base=seed)
    st = A.PassState(sc, seeds); A.run_pass(k, sc, st)
    res = {}
    for eo in (True, False):
        ctx.set_exact_only(eo)
        fr = render.FusedRenderer(ctx, sc, seeds=seeds); fr.execute_render()
        a = canon(fr.acu.read(np.float32)).ravel(); w = canon(st.acu).ravel()
        res["exact" if eo else "optimistic"] = int((a != w).sum())
        res["deferred_"+("exact" if eo else "opt")] = int(ctx.pass_deferred())
        fr.release()
    d = sc.d
    print(seed, res, "sph", d["n_spheres"], "tri", d["n_triangles"], "n_slabs", d["n_slabs"], "meshes", [(m["ntriangles"], m["nslabs"]) for m in d["meshes"]], "lights", len(d["lights"]), "rpp", d["rays_per_pixel"], flush=True)
