#!/usr/bin/env python3
"""profiles/trip_counts.py -- how often each loop of the fused pass really runs.

Needs a counting build:  MIRT_OUT=ab/count/libmirt_count.so bash 2015-raytracing_amd/csrc/build.sh -DPT_COUNT=1
Renders one frame of a scene with it and prints the wave-level trip counters (pt_trace.hpp PtCounter) per wave and per sample-lane,
next to the static VALU count of each loop body (read off the ISA: profiles/tools/isa_blocks.py), i.e. where the kernel's VALU
instructions go.  usage: MIRT_LIB_PATH=$PWD/ab/count/libmirt_count.so python profiles/trip_counts.py [scene] [width height rpp bounces]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

NAMES = ["waves", "segments", "seg_lanes", "q_closest", "q_closest_lanes", "sweep_closest", "trips_closest", "trip_lanes_closest",
         "q_shadow", "q_shadow_lanes", "sweep_shadow", "trips_shadow", "trip_lanes_shadow",
         "sph_q_closest", "sph_q_closest_lanes", "sph_tests_closest", "sph_roots_closest", "sph_q_shadow", "sph_q_shadow_lanes", "sph_tests_shadow", "sph_roots_shadow",
         "box_tests", "box_lanes", "shade", "shade_lanes", "bounce", "bounce_lanes",
         "grid_walks_closest", "grid_walks_shadow", "grid_want_lanes_closest", "grid_want_lanes_shadow", "grid_phases_closest", "grid_phases_shadow",
         "grid_a_steps_closest", "grid_a_steps_shadow", "grid_a_lane_steps_closest", "grid_a_lane_steps_shadow", "grid_rounds_closest", "grid_rounds_shadow",
         "grid_pairs_closest", "grid_pairs_shadow"]


def main():
    graft.load_package()
    from raytracing_amd.pyhost import mirt, render, scene
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    w, h, rpp, bounces = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 16, 8)))
    if name == "cornell":
        base = scene.PackedScene(open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read())
    else:
        fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        base = scene.PackedScene(bytes(fx["scene_json"]).decode())
    sc = base.resized(w, h, rpp)
    lib = mirt.lib()
    if not hasattr(lib, "mirt_debug_counters"):
        raise SystemExit("this libmirt.so is not a counting build (-DPT_COUNT=1)")
    ctx = mirt.Context(0)
    fr = render.FusedRenderer(ctx, sc, keep_acu=False)
    out = (C.c_ulonglong * 48)()
    lib.mirt_debug_counters(None, 1)
    fr.execute_render(bounces=bounces, fresh=True)
    ctx.finish()
    lib.mirt_debug_counters(out, 0)
    c = dict(zip(NAMES, list(out)))
    waves = c["waves"]
    print(json.dumps({"scene": name, "width": w, "height": h, "rpp": rpp, "bounces": bounces, "counters": c}))
    print(f"{'counter':28s} {'total':>14s} {'per wave':>10s}")
    for k in NAMES:
        print(f"{k:28s} {c[k]:14d} {c[k] / waves:10.2f}")
    fr.release()
    ctx.destroy()


if __name__ == "__main__":
    main()
