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
