"""profiles/r4_work/split_counts.py -- the walk's trip counters (a -DPT_COUNT=1 build: MIRT_LIB_PATH) on cornell_teapot3 with a mesh removed:
which mesh the walks, steps, rounds and pairs belong to.  Run on the GPU box."""
import copy, ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
from trip_counts import NAMES
fx = np.load(os.path.join(ROOT, "tests", "golden", "cornell_teapot3_32x24_r4.npz"))
d0 = json.loads(bytes(fx["scene_json"]).decode())
lib = mirt.lib()
for name in ("full", "no_teapot", "no_box"):
    d = copy.deepcopy(d0)
    if name == "no_teapot": d["meshes"] = d["meshes"][1:]
    if name == "no_box": d["meshes"] = d["meshes"][:1]
    sc = scene.PackedScene(json.dumps(d)).resized(1920, 1080, 16)
    ctx = mirt.Context(0)
    fr = render.FusedRenderer(ctx, sc, keep_acu=False)
    out = (C.c_ulonglong * 48)()
    lib.mirt_debug_counters(None, 1)
    fr.execute_render(bounces=5, fresh=True)
    ctx.finish()
    lib.mirt_debug_counters(out, 0)
    c = dict(zip(NAMES, list(out)))
    w = c["waves"]
    print(name, {k: round(c[k] / w, 1) for k in NAMES if k.startswith("grid_") or k in ("box_tests", "box_lanes")}, flush=True)
    fr.release(); ctx.destroy()
