#!/usr/bin/env python3
"""oracle/gen/gen_golden.py -- TEST INFRASTRUCTURE, build container only.

Generates tests/golden/*.npz from the REFERENCE ITSELF run here:

  inputs   the reference's own host code (Assign10 code.js + lib/ + tri/ parser)
           executed in a Node vm sandbox (ref_host_dump.js) -> every buffer it
           would upload for a scene (camera, lights, materials, grid-sorted
           spheres / triangles / meshes, cell offsets, AABBs)
  outputs  the reference's own OpenCL C kernels, compiled for x86 from
           /root/reference (oracle/Makefile `ref`) and driven through one
           progressive pass in executeRender's order (oracle/a10_pass.py)

The reference's sources are read where they lie and never copied; only the
numeric inputs/outputs land in the fixtures.  Re-run after any change to
oracle/cl_numerics.h:

    make -C oracle ref && python oracle/gen/gen_golden.py

Each fixture holds: scene_json (the packed inputs, utf-8 bytes), seeds_in,
and the buffers after the primary segment (p_*) and after the whole pass (f_*):
rays/shadow {o,d,mint,maxt}, pois {p,normal,atte,matId}, acu, seeds, pixel.
Large cases keep only per-pixel data + SHA-256 digests of the per-ray buffers.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import a10_pass as A  # noqa: E402

REFROOT = os.environ.get("REFROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

# (name, scene, width, height, rpp, full buffers?)
CASES = [
    ("basic_64x48_r1", "basic.xml", 64, 48, 1, True),
    ("basic_32x24_r4", "basic.xml", 32, 24, 4, True),
    ("triangles_64x48_r1", "triangles.xml", 64, 48, 1, True),
    ("triangles_32x24_r4", "triangles.xml", 32, 24, 4, True),
    ("cornell_64x48_r1", "cornell.xml", 64, 48, 1, True),
    ("cornell_32x24_r4", "cornell.xml", 32, 24, 4, True),
    ("cornell_16x12_r9", "cornell.xml", 16, 12, 9, True),       # odd k: lens sample (0.5,0.5) -> 0/0 in the disk map
    ("cornell_teapot3_64x48_r1", "cornell_teapot3.xml", 64, 48, 1, True),   # 2 lights, 2 meshes (n=10, n=5)
    ("cornell_teapot3_32x24_r4", "cornell_teapot3.xml", 32, 24, 4, True),
    ("cornell_official_64x48_r1", "cornell_official.xml", 64, 48, 1, True),  # dropped-on-max-face triangles
    ("twoLights_32x24_r4", "twoLights.xml", 32, 24, 4, True),
    ("threeLights_32x24_r1", "threeLights.xml", 32, 24, 1, True),
    ("basic2_32x24_r4", "basic2.xml", 32, 24, 4, True),                    # the remaining three of the reference's ten A10 scenes
    ("cornell_teapot_32x24_r4", "cornell_teapot.xml", 32, 24, 4, True),
    ("cornell_teapot2_32x24_r4", "cornell_teapot2.xml", 32, 24, 4, True),
    ("cornell_320x240_r16", "cornell.xml", 320, 240, 16, False),
    # our own scenes (tests/scenes/page): inputs packed by the REFERENCE host code, outputs by the reference kernels
    ("own_studio_48x36_r4", "@studio.xml", 48, 36, 4, True),
    ("own_gems_48x36_r4", "@gems.xml", 48, 36, 4, True),
    ("own_gems_64x48_r1", "@gems.xml", 64, 48, 1, True),
    ("own_flat_32x24_r4", "@flat.xml", 32, 24, 4, True),
]
OWN_PAGE = os.path.join(ROOT, "tests", "scenes", "page")


def host_dump(scene, w, h, rpp):
    cmd = ["node", os.path.join(HERE, "ref_host_dump.js"), REFROOT, scene.lstrip("@"), str(w), str(h), str(rpp)]
    if scene.startswith("@"):
        cmd += ["1", OWN_PAGE]
    out = subprocess.run(cmd, check=True, capture_output=True, cwd="/tmp")
    return out.stdout.decode()


def flat(prefix, snap, out):
    for k in ("rays", "shadow"):
        for f in ("o", "d", "mint", "maxt"):
            out[f"{prefix}_{k}_{f}"] = np.ascontiguousarray(snap[k][f])
    for f in ("p", "normal", "atte", "matId"):
        out[f"{prefix}_pois_{f}"] = np.ascontiguousarray(snap["pois"][f])
    out[f"{prefix}_acu"] = snap["acu"]
    out[f"{prefix}_seeds"] = snap["seeds"]


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = A.load_ref()
    only = set(sys.argv[1:])
    for name, scene, w, h, rpp, full in CASES:
        if only and name not in only:
            continue
        js = host_dump(scene, w, h, rpp)
        sc = A.Scene(json.loads(js))
        seeds = A.make_seeds(sc.total_rays)
        st = A.PassState(sc, seeds)
        ck = {}
        A.run_pass(ref, sc, st, checkpoints=ck)
        out = {"scene_json": np.frombuffer(js.encode(), dtype=np.uint8), "pixel": st.pixel,
               "radiance": A.radiance_sums(st.acu, rpp)}
        if full:
            out["seeds_in"] = seeds
            flat("p", ck["primary"], out)
            flat("f", st.snapshot(), out)
        else:
            out["f_pois_matId"] = np.ascontiguousarray(st.pois["matId"]).astype(np.int8)
            out["sha_acu"] = digest(st.acu)
            out["sha_seeds"] = digest(st.seeds)
            out["sha_rays_maxt"] = digest(st.rays["maxt"])
            out["sha_pois_atte"] = digest(st.pois["atte"])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {sc.total_rays} rays, hit {float((st.pois['matId'] >= 0).mean()):.3f}, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
