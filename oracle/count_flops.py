#!/usr/bin/env python3
"""oracle/count_flops.py -- TEST INFRASTRUCTURE.  Algorithmic fp32 operations per sample (+ - * / sqrt sin cos = 1 each,
a fused multiply-add = 2; compares, min/max, selects, conversions, integer work = 0), measured by running the CPU oracle built
with its operation counters (make -C oracle count -> liboracle_count.so, single-threaded) on a fixture's scene at 320x240 x 16
rays per pixel.  bench.py's roofline uses these constants.

    make -C oracle count && python oracle/count_flops.py [fixture ...]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import a10_pass as A  # noqa: E402

W, H, RPP = 320, 240, 16


def main():
    names = sys.argv[1:] or ["cornell_32x24_r4", "cornell_teapot3_32x24_r4"]
    k = A.CpuKernels(os.path.join(HERE, "liboracle_count.so"), "oracle_a10_")
    A.set_hw_tables(k.lib, "oracle_")
    k.lib.oracle_flops_get.restype = C.c_ulonglong
    for name in names:
        fx = np.load(os.path.join(HERE, "..", "tests", "golden", name + ".npz"))
        d = json.loads(bytes(fx["scene_json"]).decode())
        cam = list(d["cam"])
        cam[14], cam[15] = float(W), float(H)
        d.update(cam=cam, width=W, height=H, rays_per_pixel=RPP)
        sc = A.Scene(d)
        out = {}
        for bounces in (5, 8):
            st = A.PassState(sc, A.make_seeds(sc.total_rays))
            k.lib.oracle_flops_reset()
            A.run_pass(k, sc, st, bounces=bounces)
            out[bounces] = k.lib.oracle_flops_get() / sc.total_rays
        print(f"{name.rsplit('_', 2)[0]}: {out[5]:.1f} flop/sample at 5 bounces, {out[8]:.1f} at 8   ({W}x{H} x {RPP}, {sc.total_rays} samples)")


if __name__ == "__main__":
    main()
