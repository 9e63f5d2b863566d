"""profiles/cornell_split.py -- run on the GPU box: where the headline kernel's time goes.  The fused pass on variants of cornell.xml's
packed scene (a primitive set removed; the frame is no longer the reference's, only the cost structure matters), 1920x1080 x RPP rays, depth 8."""
import copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d0 = json.loads(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
ctx = mirt.Context(0)
ctx.set_profiling(True)


def variant(name):
    d = copy.deepcopy(d0)
    if name == "no_spheres":
        d["n_spheres"] = 0
    elif name == "no_triangles":
        d["n_triangles"] = 0
    return d


rpp, bounces = int(os.environ.get("RPP", "64")), int(os.environ.get("BOUNCES", "8"))
for name in os.environ.get("VARIANTS", "full,no_spheres,no_triangles").split(","):
    sc = scene.PackedScene(json.dumps(variant(name))).resized(1920, 1080, rpp)
    fr = render.FusedRenderer(ctx, sc, want_radiance=False)
    fr.execute_render(bounces=bounces)
    ms = []
    for _ in range(3):
        fr.execute_render(bounces=bounces)
        ms.append(ctx.pass_timing()[0])
    print(json.dumps({"variant": name, "rpp": rpp, "bounces": bounces, "kernel_ms": round(sum(ms) / len(ms), 3)}), flush=True)
    fr.release()
