"""profiles/tile_balance.py -- run on the GPU box (one GPU): how evenly the headline frame splits into the N contiguous row tiles
bench.py gives N ranks.  Renders every tile of N = 2, 4, 8 on this GPU and prints the fused kernel's time per tile; the N-GPU frame
takes max(tile) while perfect balance would take mean(tile), so mean/max bounds the strong-scaling efficiency from above."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene, tiling
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
bounces = int(os.environ.get("BOUNCES", "8"))
ctx = mirt.Context(0)
ctx.set_profiling(True)
for world in (1, 2, 4, 8):
    ms = []
    for row0, nrows in tiling.row_tiles(sc.height, world):
        fr = render.FusedRenderer(ctx, sc, row0=row0, nrows=nrows, want_radiance=False)
        fr.execute_render(bounces=bounces)
        ctx.zero(fr.acu); ctx.seed_fill(fr.seeds, fr.first_ray, fr.nrays, 0); fr.passes = 1
        fr.execute_render(bounces=bounces)
        ms.append(ctx.pass_timing()[0])
        fr.release()
    print(json.dumps({"tiles": world, "kernel_ms": [round(m, 2) for m in ms], "max": round(max(ms), 2), "mean": round(float(np.mean(ms)), 2),
                      "balance": round(float(np.mean(ms)) / max(ms), 4)}), flush=True)
