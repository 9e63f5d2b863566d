"""profiles/contract_drift.py -- run on the GPU box.  How far do two CONFORMANT OpenCL builds of the reference drift apart?
Both are the reference's code.cl compiled by AMD's OpenCL toolchain for gfx950 (oracle/Makefile ref_gpu), run on the device:
  A  -cl-fp32-correctly-rounded-divide-sqrt   (the contract: IEEE / and sqrt, reproducible on a CPU)
  B  default options                           (AMD's 2.5-ulp v_rcp_f32 division, 3-ulp sqrt)
Same scene, same seeds (cornell.xml 320x240 x 16 rays, one pass, five bounces).  Reports, after the primary segment and after the whole
pass: the fraction of rays whose hit id (matId) differs, the RMS difference of the per-pixel radiance (sum of its 16 accumulators / 16),
and the largest difference of an 8-bit channel of the resolved frame.  This is what "within 1e-5 RMS of the reference OpenCL output"
can and cannot mean: a path tracer is chaotic, one ulp in a division flips a hit five bounces later."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import a10_pass as A, ref_gpu as G

def run(hsaco, sc, seeds):
    k = G.GpuRefKernels(hsaco)
    st = A.PassState(sc, seeds); ck = {}
    A.run_pass(k, sc, st, checkpoints=ck)
    k.release()
    return ck["primary"], st

for name in ("cornell_320x240_r16", "cornell_teapot3_32x24_r4"):
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    d = json.loads(bytes(fx["scene_json"]).decode())
    if name.startswith("cornell_teapot3"):
        cam = list(d["cam"]); cam[14], cam[15] = 320.0, 240.0; cam[12] = float(np.float32(cam[13] * 320 / 240))
        d.update(cam=cam, width=320, height=240, rays_per_pixel=16)
    sc = A.Scene(d)
    seeds = A.make_seeds(sc.total_rays)
    pa, a = run(os.path.join(ROOT, "oracle", "_ref", "a10_gfx950.hsaco"), sc, seeds)
    pb, b = run(os.path.join(ROOT, "oracle", "_ref", "a10_gfx950_default.hsaco"), sc, seeds)
    ra, rb = A.radiance_sums(a.acu, sc.rpp)[:, :3] / sc.rpp, A.radiance_sums(b.acu, sc.rpp)[:, :3] / sc.rpp
    out = {"scene": name.rsplit("_", 2)[0], "rays": sc.total_rays,
           "hit_id_flips_primary": float((pa["pois"]["matId"] != pb["pois"]["matId"]).mean()),
           "hit_id_flips_after_pass": float((a.pois["matId"] != b.pois["matId"]).mean()),
           "radiance_rms_per_pixel": float(np.sqrt(((ra - rb).astype(np.float64) ** 2).mean())),
           "radiance_mean": float(ra.mean()),
           "max_8bit_channel_diff": int(np.abs(a.pixel.astype(int) - b.pixel.astype(int)).max()),
           "mean_frame_diff_8bit": float(np.abs(a.pixel[:, :3].astype(float).mean(axis=0) - b.pixel[:, :3].astype(float).mean(axis=0)).max()),
           "seeds_equal": bool(np.array_equal(a.seeds, b.seeds))}
    print(json.dumps(out), flush=True)
