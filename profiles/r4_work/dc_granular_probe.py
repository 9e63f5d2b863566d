import os, sys, json, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"oracle")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import __graft_entry__ as graft
import a10_pass as A, ref_gpu as G
from conftest import canon, load_fixture
graft.load_package()
from raytracing_amd.pyhost import mirt, render, scene
hs = os.path.join(ROOT,"oracle","_ref","a10_gfx950_default.hsaco" if os.environ.get("MIRT_CONTRACT")=="default" else "a10_gfx950.hsaco")
print("reference:", hs)
for name in ("basic_32x24_r4","cornell_32x24_r4","triangles_32x24_r4","twoLights_32x24_r4","cornell_teapot3_32x24_r4","own_gems_48x36_r4","cornell_official_64x48_r1"):
    fx, sc0 = load_fixture(name)
    ps = scene.PackedScene(dict(sc0.d)).resized(96, 54, 4)
    sc = A.Scene(ps.d)
    seeds = A.make_seeds(sc.total_rays, seed_base=5)
    k = G.GpuRefKernels(hs)
    st = A.PassState(sc, seeds); A.run_pass(k, sc, st)
    ctx = mirt.Context(0); ctx.set_fusion(0)
    gr = render.GranularRenderer(ctx, ps, seeds=seeds); gr.execute_render()
    sh = gr.read("shadow").view(A.RAY_DT); ry = gr.read("rays").view(A.RAY_DT)
    res = {}
    for tag, g_, w_ in (("rays", ry, st.rays), ("shadow", sh, st.shadow)):
        for f in ("mint","maxt"):
            a, b = canon(np.ascontiguousarray(g_[f])), canon(np.ascontiguousarray(w_[f]))
            res[tag+"."+f] = int((a != b).sum())
    res["acu"] = int((canon(gr.read("acu")) != canon(st.acu).ravel()).sum())
    bad = np.flatnonzero(canon(np.ascontiguousarray(sh["mint"])) != canon(np.ascontiguousarray(st.shadow["mint"])))
    ex = [(int(i), float(sh["mint"][i]), float(st.shadow["mint"][i]), float(sh["maxt"][i]), float(st.shadow["maxt"][i])) for i in bad[:3]]
    print(name, res, ex, "sph", sc.d["n_spheres"], "tri", sc.d["n_triangles"], "meshes", [(m["ntriangles"], m["nslabs"]) for m in sc.d["meshes"]], flush=True)
    k.release(); gr.release(); ctx.destroy()
