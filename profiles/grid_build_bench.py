"""profiles/grid_build_bench.py -- run on the GPU box.  SURVEY 8f rank 1's stated reason for a device grid builder was "the host is the
bottleneck for house_of_parliament at large n_slabs": time mirt_grid_build (+ the triangle gather) against our JavaScript host's
binning (host/scene.js buildGrid + slot copy, a counting sort; the reference's nested JS arrays are slower still) on that mesh's
9 144 triangles (they ride in the Assign04 fixture) at n = 2, 16, 32, 64."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt
from test_grid_build import frame_job, unique_triangles
flat = frame_job("frame_a04_parliament_96x64")
tri = unique_triangles(flat)
b = np.asarray(flat["bounds"], np.float64); b6 = [b[0], b[1], b[2], b[4], b[5], b[6]]
ctx = mirt.Context(0)
with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
    json.dump({"tri": tri.ravel().tolist(), "b6": b6}, f)
    path = f.name
js = """
const s = require(process.argv[1]); const fs = require('fs');
const d = JSON.parse(fs.readFileSync(process.argv[2], 'utf8')); const P = d.tri, n = +process.argv[3];
const B = new s.Bounds(d.b6.slice(0,3), d.b6.slice(3));
const box = (i) => { const o = 9*i, mn=(a,b,c)=>Math.min(Math.min(a,b),c), mx=(a,b,c)=>Math.max(Math.max(a,b),c);
  return [[mn(P[o],P[o+3],P[o+6]),mn(P[o+1],P[o+4],P[o+7]),mn(P[o+2],P[o+5],P[o+8])],[mx(P[o],P[o+3],P[o+6]),mx(P[o+1],P[o+4],P[o+7]),mx(P[o+2],P[o+5],P[o+8])]]; };
let best = 1e9, slots = 0;
for (let r = 0; r < 5; r++) { const t0 = process.hrtime.bigint(); const g = s.buildGrid(P.length/9, n, B, box);
  const pos = new Float32Array(g.order.length*12); for (let k = 0; k < g.order.length; k++) for (let v = 0; v < 3; v++) for (let c = 0; c < 3; c++) pos[12*k+4*v+c] = P[9*g.order[k]+3*v+c];
  const ms = Number(process.hrtime.bigint() - t0) / 1e6; best = Math.min(best, ms); slots = g.order.length; }
console.log(JSON.stringify({ms: best, slots: slots}));
"""
for n in (2, 16, 32, 64):
    best = 1e9
    for r in range(6):
        ctx.finish(); t0 = time.perf_counter()
        off, order, total = ctx.grid_build(1, tri, b6, n)
        pos, _ = ctx.grid_gather_triangles(order, total, tri)
        ctx.finish(); dt = (time.perf_counter() - t0) * 1e3
        for x in (off, order, pos): x.release()
        if r: best = min(best, dt)
    r = json.loads(subprocess.run(["node", "-e", js, os.path.join(ROOT, "2015-raytracing_amd", "host", "scene.js"), path, str(n)], capture_output=True, check=True).stdout)
    print(json.dumps({"n_slabs": n, "slots": total, "device_ms_incl_upload": round(best, 3), "js_ms": round(r["ms"], 3), "js_slots": r["slots"]}), flush=True)
ctx.destroy()
