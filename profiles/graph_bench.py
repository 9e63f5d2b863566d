"""profiles/graph_bench.py -- run on the GPU box: the reference-shaped pass at the page's own defaults (320x240 canvas, index.html:46;
one ray per pixel, code.js:400), enqueued kernel by kernel vs replayed as one HIP graph."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = mirt.Context(0)
for name in ("cornell_64x48_r1", "cornell_teapot3_64x48_r1"):
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    for w, h, rpp in ((320, 240, 1), (320, 240, 4), (1920, 1080, 1)):
        sc = scene.PackedScene(bytes(fx["scene_json"]).decode()).resized(w, h, rpp)
        row = {"scene": name.rsplit("_", 2)[0], "size": f"{w}x{h}", "rpp": rpp}
        for mode in (False, True):
            gr = render.GranularRenderer(ctx, sc)
            for _ in range(3):
                gr.execute_render(use_graph=mode)
            n = 50
            t0 = time.perf_counter()
            for _ in range(n):
                gr.execute_render(use_graph=mode)
            row["graph_ms" if mode else "enqueue_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
            gr.release()
        print(json.dumps(row), flush=True)
