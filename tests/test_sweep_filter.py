"""The candidate sweep's plane window (csrc/pt_trace.hpp trace_cell1, LANES) drops a triangle only when the reference's own test rejects
it -- checked on the CPU, without a GPU: oracle/sweep_check.c restates the sweep (the constants of k_planeRuns included) beside the
reference's interTriangle (A10 code.cl:250-288) under the numerics contract and counts violations over random and adversarial cases
(rays aimed at the triangle, window edges on the reference's own t and a few ulps either side, forty octaves of scale, needles, rays nearly
in the plane).  The GPU tests establish the same end to end (every frame bit-identical to the reference binary); this isolates the claim."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def check():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.oracle_sweep_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    lib.oracle_sweep_check.restype = None

    def run(seed, count, shrink=0):
        out = (C.c_uint64 * 6)()
        lib.oracle_sweep_check(seed, count, shrink, out)
        return dict(zip(("cases", "violations", "accepted", "rejected", "dropped", "skipped"), out))
    return run


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_the_sweep_never_drops_what_the_reference_accepts(check, seed):
    r = check(seed, 25_000_000)
    assert r["cases"] > 20_000_000 and r["accepted"] > 500_000, r       # the adversarial windows do produce accepted hits on their edges
    assert r["violations"] == 0, r
    assert r["dropped"] > 0.75 * r["rejected"], r                        # ... and it is a filter: most of what the reference rejects never reaches a test


def test_the_check_sees_a_margin_that_is_too_small(check):
    """The product's margin is 128 u E (|o|_1 + |p0|_1), nine times the bound of pt_trace.hpp's derivation.  Cut to an eighth of a rounding
    unit the sweep does drop hits: the check can tell."""
    assert check(7, 20_000_000, shrink=10)["violations"] > 0
    assert check(7, 20_000_000, shrink=8)["violations"] == 0             # half a unit still holds on this sample: the distance is roundings, not luck
