import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = np.load(os.path.join(ROOT, "tests", "golden", "cornell_64x48_r1.npz"))
sc = scene.PackedScene(bytes(fx["scene_json"]).decode())
ctx = mirt.Context(0)
gr = render.GranularRenderer(ctx, sc)
for p in range(5):
    t0 = time.perf_counter(); gr.execute_render(use_graph=True); print("pass", p, round(time.perf_counter() - t0, 3), "s", flush=True)
t0 = time.perf_counter(); ctx.capture_begin()
try:
    ctx.finish()
except Exception as e:
    print("refused:", str(e)[:60])
gph = ctx.capture_end(); print("empty capture end", round(time.perf_counter() - t0, 3)); 
t0 = time.perf_counter(); ctx.graph_release(gph); gr.release(); print("release", round(time.perf_counter() - t0, 3))
