#!/usr/bin/env python3
"""oracle/count_flops_frames.py -- TEST INFRASTRUCTURE.  Algorithmic fp32 operations per PIXEL of BASELINE configs 2 and 3 (SURVEY.md 8d:
"flops = W*H*sum(tests by exit)"; + - * / sqrt = 1 each, a fused multiply-add = 2), measured by running the CPU oracle built with its
operation counters (make -C oracle count) on the frame jobs the fixtures carry, at the configs' full sizes: Assign04 brute force on
house_of_parliament (9 144 triangles) and teapot (992) at 1024 x 1024; Assign07 grid on house_of_parliament at 1920 x 1080, n_slabs 2 / 16 / 32.
bench.py's `frames` record uses these constants (FRAME_FLOPS_PER_PIXEL).

    make -C oracle count && python oracle/count_flops_frames.py
"""
import ctypes as C
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import frame_pass as F  # noqa: E402
from test_frames import fixture, regrid, resized  # noqa: E402


def main():
    _, a04 = fixture("frame_a04_parliament_96x64")
    _, a07 = fixture("frame_a07_parliament_n16_160x120")
    _, tea = fixture("frame_a04_teapot_160x120")
    lib, _ = F._lib("count", 4)
    lib.oracle_flops_get.restype = C.c_ulonglong
    jobs = [("a04_teapot_1024", resized(tea, 1024, 1024)), ("a04_parliament_1024", resized(a04, 1024, 1024))]
    jobs += [(f"a07_parliament_1080p_n{n}", resized(a07 if n == 16 else regrid(a07, a04, n), 1920, 1080)) for n in (2, 16, 32)]
    for tag, d in jobs:
        lib.oracle_flops_reset()
        t0 = time.time()
        F.run_frame("count", F.Frame(d))
        fl = lib.oracle_flops_get()
        print(f"{tag}: {fl} flop per frame, {fl / (d['width'] * d['height']):.1f} per pixel ({time.time() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
