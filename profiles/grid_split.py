"""profiles/grid_split.py -- run on the GPU box: where cornell_teapot3's time goes.  The fused pass on variants of the packed scene
(a mesh or a light removed; the frame is no longer the reference's, only the cost structure matters here), 1920x1080 x 16 rays."""
import copy, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = np.load(os.path.join(ROOT, "tests", "golden", "cornell_teapot3_32x24_r4.npz"))
d0 = json.loads(bytes(fx["scene_json"]).decode())
ctx = mirt.Context(0)


def variant(name):
    d = copy.deepcopy(d0)
    if name == "no_teapot":
        d["meshes"] = d["meshes"][1:]
    elif name == "no_box":
        d["meshes"] = d["meshes"][:1]
    elif name == "no_meshes":
        d["meshes"] = []
    elif name == "one_light":
        d["lights"] = d["lights"][:1]
    elif name == "no_loose":
        d["n_spheres"] = 0; d["n_triangles"] = 0
    return d


for name in os.environ.get("VARIANTS", "full,no_teapot,no_box,no_meshes,one_light").split(","):
    sc = scene.PackedScene(json.dumps(variant(name))).resized(1920, 1080, int(os.environ.get("RPP", "16")))
    fr = render.FusedRenderer(ctx, sc, want_radiance=False)
    fr.execute_render()
    ctx.finish()
    t0 = time.perf_counter()
    for _ in range(3):
        fr.execute_render()
    ctx.finish()
    print(json.dumps({"variant": name, "ms_per_pass": round((time.perf_counter() - t0) / 3 * 1e3, 2)}), flush=True)
    fr.release()
