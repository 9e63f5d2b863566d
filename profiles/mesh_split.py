"""profiles/mesh_split.py -- run on the GPU box: cornell_teapot3 at 1080p x 16 with both meshes, each one alone, and none; how much of
the frame is spent in which grid walk (and what walking them at the same time could save at most)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = np.load(os.path.join(ROOT, "tests", "golden", "cornell_teapot3_32x24_r4.npz"))
d0 = json.loads(bytes(fx["scene_json"]).decode())
ctx = mirt.Context(0)
for tag, keep in (("both", [0, 1]), ("teapot only", [0]), ("box only", [1]), ("no mesh", [])):
    d = dict(d0); d["meshes"] = [d0["meshes"][i] for i in keep]
    sc = scene.PackedScene(json.dumps(d)).resized(1920, 1080, 16)
    fr = render.FusedRenderer(ctx, sc, want_radiance=False)
    fr.execute_render(); ctx.finish()
    t0 = time.perf_counter()
    for _ in range(3):
        fr.execute_render()
    ctx.finish()
    print(tag, "meshes", [(m["ntriangles"], m["nslabs"]) for m in d["meshes"]], round((time.perf_counter() - t0) / 3 * 1e3, 2), "ms", flush=True)
    fr.release()
