"""profiles/scene_bench.py -- run on the GPU box: the fused pass on every scene the fixtures carry (the reference's A10 scenes and
this repo's own), re-sized to 1920x1080 at 16 rays per pixel; prints ms per pass, Msamples/s and how many samples the optimistic
kernel handed to the exact one.  Scene packs come from tests/golden/*.npz (data written by the reference host, see oracle/gen)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["basic_32x24_r4", "cornell_32x24_r4", "triangles_32x24_r4", "twoLights_32x24_r4", "threeLights_32x24_r1", "cornell_official_64x48_r1",
         "cornell_teapot3_32x24_r4", "own_flat_32x24_r4", "own_gems_48x36_r4", "own_studio_48x36_r4"]
if os.environ.get("SCENES"):
    CASES = os.environ["SCENES"].split(",")
ctx = mirt.Context(0)
if os.environ.get("MIRT_EXACT_ONLY") == "1":
    ctx.set_exact_only(True)
rpp = int(os.environ.get("RPP", "16"))
for name in CASES:
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    base = scene.PackedScene(bytes(fx["scene_json"]).decode())
    sc = base.resized(1920, 1080, rpp)
    fr = render.FusedRenderer(ctx, sc, want_radiance=False)
    fr.execute_render()
    ctx.finish()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        fr.execute_render()
    ctx.finish()
    dt = (time.perf_counter() - t0) / n
    d = sc.d
    print(json.dumps({"scene": name.rsplit("_", 2)[0], "spheres": d["n_spheres"], "triangles": d["n_triangles"], "meshes": [[m["ntriangles"], m["nslabs"]] for m in d["meshes"]],
                      "n_slabs": d["n_slabs"], "lights": len(d["lights"]), "ms_per_pass": round(dt * 1e3, 2), "Msamples_s": round(sc.total_rays / dt / 1e6, 1),
                      "deferred": int(ctx.pass_deferred())}), flush=True)
    fr.release()
