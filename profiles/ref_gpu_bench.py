"""profiles/ref_gpu_bench.py -- run on the GPU box.  The REFERENCE'S OWN kernels (its code.cl as AMD's OpenCL toolchain compiles it for gfx950,
oracle/_ref/a10_gfx950.hsaco), driven through executeRender's launch sequence on the MI355X, timed: what the reference itself achieves on
this GPU, kernel by kernel, beside our kernel-by-kernel path and our fused pass on the same frame.  cornell.xml 1920x1080, depth 5 and 8."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render, scene
import a10_pass as A, ref_gpu as G
# SCENE=<fixture name> (e.g. cornell_teapot3_32x24_r4): another A10 scene the fixtures carry, re-sized; default the headline scene
if os.environ.get("SCENE"):
    base = scene.PackedScene(bytes(np.load(os.path.join(ROOT, "tests", "golden", os.environ["SCENE"] + ".npz"))["scene_json"]).decode())
else:
    base = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
ctx = mirt.Context(0)
k = G.GpuRefKernels()
for rpp in [int(x) for x in os.environ.get("RPPS", "16,64").split(",")]:
    ps = base.resized(1920, 1080, rpp)
    if not os.environ.get("SCENE"):
        ps.cam = base.cam.copy()
    sc = A.Scene(ps.d)
    n, npix = sc.total_rays, sc.width * sc.height
    class St: pass
    st = St()
    st.rays, st.pois, st.shadow = G.DevBuf(n * 48), G.DevBuf(n * 64), G.DevBuf(n * 48)
    st.acu, st.seeds, st.pixel = G.DevBuf(n * 16), G.DevBuf(n * 4), G.DevBuf(npix * 4)
    st.passes = 1
    st.seeds.upload(A.make_seeds(n))
    for bounces in (5, 8):
        A.run_pass(k, sc, st, bounces=bounces)                  # warm-up
        G.chk(G.hip().hipDeviceSynchronize(), "sync")
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            A.run_pass(k, sc, st, bounces=bounces, init_acu=False)
        G.chk(G.hip().hipDeviceSynchronize(), "sync")
        dt = (time.perf_counter() - t0) / reps
        gr = render.GranularRenderer(ctx, ps)                   # our kernel-by-kernel path, same launch structure
        gr.execute_render(bounces=bounces); ctx.finish()
        t2 = time.perf_counter()
        for _ in range(reps): gr.execute_render(bounces=bounces)
        ctx.finish()
        dg = (time.perf_counter() - t2) / reps
        gr.release()
        fr = render.FusedRenderer(ctx, ps, want_radiance=False)
        fr.execute_render(bounces=bounces); ctx.finish()
        t1 = time.perf_counter()
        for _ in range(reps): fr.execute_render(bounces=bounces)
        ctx.finish()
        df = (time.perf_counter() - t1) / reps
        fr.release()
        print(json.dumps({"rpp": rpp, "bounces": bounces, "samples": n, "reference_kernels_ms_per_pass": round(dt * 1e3, 2), "reference_Msamples_s": round(n / dt / 1e6, 1),
                          "mirt_kernel_by_kernel_ms_per_pass": round(dg * 1e3, 2), "mirt_fused_ms_per_pass": round(df * 1e3, 2), "mirt_fused_Msamples_s": round(n / df / 1e6, 1), "speedup": round(dt / df, 2)}), flush=True)
    for b in (st.rays, st.pois, st.shadow, st.acu, st.seeds, st.pixel): b.free()
k.release(); ctx.destroy()
