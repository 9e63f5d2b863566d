#!/usr/bin/env python3
"""tests/grid_waves_check.py -- run by tests/test_gpu_parity.py in a process of its own with MIRT_GRID_WAVES=5: the occupancy variant of the grid
kernels (k_fusedPass<true, 1, 5>: 96 registers, five waves per SIMD, no scratch -- what launch_fused picks by itself only for scenes whose cell
tables leave room for five blocks per CU) on the fixtures with grid meshes, against the compiled reference's outputs, tolerance 0.  The switch is
read once per process, hence the process."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
from conftest import bits, load_fixture  # noqa: E402


def main():
    assert os.environ.get("MIRT_GRID_WAVES") == "5"
    graft.load_package()
    from raytracing_amd.pyhost import mirt, render
    ctx = mirt.Context(0)
    ok = True
    for name in ("cornell_teapot3_32x24_r4", "cornell_teapot_32x24_r4", "cornell_teapot2_32x24_r4", "own_gems_48x36_r4", "cornell_official_64x48_r1"):
        fx, sc = load_fixture(name)
        for keep_acu in (True, False):
            fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"], keep_acu=keep_acu)
            fr.execute_render(fresh=True)
            same = np.array_equal(fr.seeds.read(np.int32), fx["f_seeds"]) and np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), fx["pixel"])
            if keep_acu:
                same = same and np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(fx["f_acu"]))
            fr.release()
            print(json.dumps({"scene": name, "keep_acu": keep_acu, "ok": bool(same)}), flush=True)
            ok = ok and same
    ctx.destroy()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
