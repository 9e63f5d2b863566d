"""profiles/pass_overhead.py -- run on the GPU box: wall time per progressive pass at small rays-per-pixel (where a pass is short and
fixed per-pass costs show), default optimistic pair vs the single exact kernel."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
ctx = mirt.Context(0)
for rpp in (1, 4, 16):
    sc = base.resized(1920, 1080, rpp)
    sc.cam = base.cam.copy()
    row = {"rpp": rpp, "samples": sc.total_rays}
    for mode in ("optimistic", "exact_only"):
        ctx.set_exact_only(mode == "exact_only")
        fr = render.FusedRenderer(ctx, sc, want_radiance=False)
        fr.execute_render(); ctx.finish()
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            fr.execute_render()
        ctx.finish()
        row[mode + "_ms_per_pass"] = round((time.perf_counter() - t0) / n * 1e3, 3)
        fr.release()
    ctx.set_exact_only(False)
    print(json.dumps(row), flush=True)
