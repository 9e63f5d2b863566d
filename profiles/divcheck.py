"""profiles/divcheck.py -- run on the GPU box: exhaustive / randomized device-side checks of the shared-reciprocal
division forms (pt_numerics.hpp) against the compiler's correctly rounded division."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from raytracing_amd.pyhost import mirt
c = mirt.Context(0)
r = c.divcheck(3, 0, 1 << 32)
print("all 2^32 denominators: rcp_refined != 1/d for", int(r[1]), "patterns; |d| bits range", hex(int(r[10])), hex(int(r[11])))
for mode, count in ((2, 1 << 38), (0, 1 << 32), (1, 81 << 23)):
    r = c.divcheck(mode, 777 + mode, count)
    print("mode", mode, "pairs", count, "| 5-op quotient", int(r[0]), "| 3-op reciprocal", int(r[1]), "| 5-op reciprocal", int(r[2]), "| 3-op quotient", int(r[3]),
          "| first mismatch (n,d)", hex(int(r[4])), hex(int(r[5])))

if "--exhaustive" in sys.argv:
    # all 2^23 x 2^23 mantissa pairs of div_exact3 against n/d, 2^16 denominators per launch; a progress line every 8 launches
    import time
    bad, t0, step = 0, time.time(), 1 << 16
    for k, first in enumerate(range(0, 1 << 23, step)):
        r = c.divcheck(4, first, step)
        bad += int(r[3])
        if int(r[3]):
            print("MISMATCH dm block", hex(first), "count", int(r[3]), "first (n,d) bits", hex(int(r[6])), hex(int(r[7])), flush=True)
        if k % 8 == 7:
            print(f"exhaustive: {first + step:#x} of 0x800000 denominators x 2^23 numerators, mismatches so far {bad}, {time.time() - t0:.0f} s", flush=True)
    print(f"EXHAUSTIVE div_exact3: 2^46 mantissa pairs, {bad} mismatches, {time.time() - t0:.0f} s")
