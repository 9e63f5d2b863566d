"""profiles/frame_bench.py -- run on the GPU box: BASELINE configs 2 and 3 at full size (house_of_parliament, 9 144 triangles: Assign04 brute force
1024 x 1024; Assign07 grid 1920 x 1080 at n = 2 / 16 / 32) and Assign04 on teapot.json, kernel time from the context's HIP-event timer."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt, render
from test_frames import fixture, regrid, resized
ctx = mirt.Context(0)


def run(tag, d):
    fp = render.FramePacked(d)
    render.render_frame(ctx, fp)            # warm-up (prepares the triangles)
    ms = []
    for _ in range(5):
        ctx.timer_start()
        render.render_frame(ctx, fp)
        ms.append(ctx.timer_stop_ms())
    print(json.dumps({"case": tag, "frame_ms_min": round(min(ms), 3), "frame_ms_median": round(float(np.median(ms)), 3)}), flush=True)


_, a04 = fixture("frame_a04_parliament_96x64")
_, a07 = fixture("frame_a07_parliament_n16_160x120")
_, tea = fixture("frame_a04_teapot_160x120")
run("a04_parliament_1024", resized(a04, 1024, 1024))
run("a04_teapot_1024", resized(tea, 1024, 1024))
for n in (2, 16, 32):
    run(f"a07_parliament_1080p_n{n}", resized(a07 if n == 16 else regrid(a07, a04, n), 1920, 1080))
