#!/usr/bin/env python3
"""profiles/walk_stats.py [fixture ...] -- what the grid walks of a scene consist of (CPU, the counting oracle: make -C oracle count): walks, cells,
tests, and how many tests repeat a triangle the same ray already tested in an earlier cell of the walk -- split by why the earlier test failed
(a reason no cell changes: facing / barycentrics; or the cell's window alone).  VERDICT r3 item 1c: a per-ray "already rejected" filter is only
worth building if the repeats are a large share of the tests.  Scene at 320x240 x 16 rays, five bounces."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import a10_pass as A  # noqa: E402

NAMES = ["walks", "cells", "cells_with_a_list", "tests", "repeat_tests", "repeat_after_fixed_reject", "repeat_after_window_reject", "walks_with_a_hit",
         "tests_beyond_the_rays_end"]


def main():
    names = sys.argv[1:] or ["cornell_teapot3_32x24_r4", "cornell_teapot_32x24_r4", "own_gems_48x36_r4"]
    k = A.CpuKernels(os.path.join(ROOT, "oracle", "liboracle_count.so"), "oracle_a10_")
    A.set_hw_tables(k.lib, "oracle_")
    for name in names:
        fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        d = json.loads(bytes(fx["scene_json"]).decode())
        cam = list(d["cam"])
        cam[14], cam[15] = 320.0, 240.0
        d.update(cam=cam, width=320, height=240, rays_per_pixel=16)
        sc = A.Scene(d)
        st = A.PassState(sc, A.make_seeds(sc.total_rays))
        k.lib.oracle_walk_stats_reset()
        A.run_pass(k, sc, st, bounces=5)
        out = (C.c_ulonglong * 16)()
        k.lib.oracle_walk_stats_get(out)
        c = dict(zip(NAMES, list(out)))
        t = max(c["tests"], 1)
        print(json.dumps({"scene": name.rsplit("_", 2)[0], "samples": sc.total_rays, **c,
                          "repeat_share": round(c["repeat_tests"] / t, 4), "repeat_after_fixed_reject_share": round(c["repeat_after_fixed_reject"] / t, 4),
                          "tests_per_walk": round(t / max(c["walks"], 1), 2), "cells_per_walk": round(c["cells"] / max(c["walks"], 1), 2)}))


if __name__ == "__main__":
    main()
