"""profiles/granular_bench.py -- run on the GPU box: the reference-shaped path (fourteen kernels enqueued one by one, as the
reference host does) on cornell.xml 1920x1080 at 16 and 64 rays per pixel; prints ms per pass, Msamples/s and the HBM rate
implied by the algorithmic 4.2 KB/sample of that path.  This path is HBM-bound by construction."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
ctx = mirt.Context(0)
# per sample: primary gen 84 B + lightRender 48 B + 6 segments x 568 B + 5 bounces x 128 B + resolve 20 B  
BYTES = 84 + 48 + 6 * 568 + 5 * 128 + 20
for rpp in [int(x) for x in os.environ.get("RPPS", "16,64").split(",")]:
    sc = base.resized(1920, 1080, rpp)
    sc.cam = base.cam.copy()
    gr = render.GranularRenderer(ctx, sc)
    gr.execute_render()            # warm-up pass
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        gr.execute_render()
    ctx.finish()
    dt = (time.perf_counter() - t0) / n
    # the same enqueues with command-stream fusion on (mirt_ctx_set_fusion(ctx, 2)): the runtime recognises executeRender's stream
    ctx.set_fusion(2)
    f0 = ctx.fused_passes()
    gf = render.GranularRenderer(ctx, sc)
    gf.execute_render()
    t2 = time.perf_counter()
    for _ in range(n):
        gf.execute_render()
    ctx.finish()
    dg = (time.perf_counter() - t2) / n
    assert ctx.fused_passes() - f0 == n + 1
    ctx.set_fusion(0)
    gf.release()
    fr = render.FusedRenderer(ctx, sc, want_radiance=False)
    fr.execute_render()
    ctx.finish()
    t1 = time.perf_counter()
    for _ in range(n):
        fr.execute_render()
    ctx.finish()
    df = (time.perf_counter() - t1) / n
    print(json.dumps({"rpp": rpp, "samples": sc.total_rays, "granular_ms_per_pass": round(dt * 1e3, 2), "granular_Msamples_s": round(sc.total_rays / dt / 1e6, 1),
                      "granular_algorithmic_GBs": round(BYTES * sc.total_rays / dt / 1e9, 1), "granular_stream_fused_ms_per_pass": round(dg * 1e3, 2), "granular_stream_fused_Msamples_s": round(sc.total_rays / dg / 1e6, 1),
                      "fused_ms_per_pass": round(df * 1e3, 2),
                      "fused_over_granular": round(dt / df, 2)}))
    gr.release(); fr.release()
