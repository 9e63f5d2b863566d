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


# ---- the layout the kernel consumes: k_planeList restated (oracle_plane_list) and the sweep's walk over it (oracle_sweep_words) ----

import numpy as np  # noqa: E402


def _lib():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.oracle_plane_list.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.oracle_plane_list.restype = C.c_uint32
    lib.oracle_sweep_words.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    lib.oracle_sweep_words.restype = None
    lib.oracle_sweep_ref_accepts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float]
    lib.oracle_sweep_ref_accepts.restype = C.c_int
    return lib


def soup(rng, n, shuffle):
    """n prepared records {p0, n.x}{e1, n.y}{e2, n.z}: rectangles cut in two (axis-aligned on a 1/64 lattice, or anywhere), some of them
    double-sided (the same two triangles again with the winding reversed); shuffle: the halves and the twins end up anywhere in the list."""
    tris = []
    while len(tris) < n:
        c = rng.uniform(-0.8, 0.8, size=3)
        if rng.random() < 0.6:
            ax = int(rng.integers(0, 3))
            u, w = np.zeros(3), np.zeros(3)
            u[(ax + 1) % 3] = float(rng.integers(2, 40)) / 64.0
            w[(ax + 2) % 3] = float(rng.integers(2, 40)) / 64.0
            c = np.round(c * 64.0) / 64.0
            if rng.random() < 0.5:
                u, w = w, u
        else:
            u, w = rng.normal(size=3) * 0.3, rng.normal(size=3) * 0.3
        p = [c, c + u, c + u + w, c + w]
        quad = [(p[0], p[1], p[2]), (p[0], p[2], p[3])]
        if rng.random() < 0.3:
            quad += [(a, cc, b) for a, b, cc in quad]      # the back faces
        tris += quad
    tris = tris[:n]
    if shuffle:
        tris = [tris[i] for i in rng.permutation(n)]
    prep = np.zeros((n, 12), np.float32)
    for i, (a, b, cc) in enumerate(tris):
        p0, p1, p2 = (np.asarray(x, np.float32) for x in (a, b, cc))
        e1, e2 = p1 - p0, p2 - p0
        nn = np.cross(e2.astype(np.float64), e1.astype(np.float64)).astype(np.float32)
        prep[i] = [p0[0], p0[1], p0[2], nn[0], e1[0], e1[1], e1[2], nn[1], e2[0], e2[1], e2[2], nn[2]]
    return prep


def parse_list(words, count):
    """(chunk, class, mask, back, entry words) of every non-padding entry"""
    out = []
    for c in range((count + 31) // 32):
        groups, first = int(words[c]), int(words[4 + c])
        g = 16 + 16 * first
        for cl in range(4):
            per, nw = (4, 4) if cl < 3 else (2, 8)
            for _ in range((groups >> (8 * cl)) & 255):
                for k in range(per):
                    e = words[g + nw * k: g + nw * (k + 1)]
                    mask, back = (int(e[2]), int(e[3])) if cl < 3 else (int(e[4]), int(e[5]))
                    if mask | back:
                        out.append((c, cl, mask, back, e.copy()))
                g += 16
    return out


@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("n", [1, 2, 3, 12, 31, 32, 33, 63, 64, 65, 95, 96])
def test_plane_list_layout(n, shuffle):
    lib = _lib()
    rng = np.random.default_rng(1000 * n + shuffle)
    prep = soup(rng, n, shuffle)
    words = np.zeros(16 + 4 * 320, np.uint32)
    used = lib.oracle_plane_list(prep.ctypes.data, n, words.ctypes.data)
    assert used <= words.size and (used - 16) % 16 == 0
    ents = parse_list(words, n)
    # every record in exactly one mask of its own chunk, no bit beyond the chunk's records
    for c in range((n + 31) // 32):
        cnt = min(32, n - 32 * c)
        seen = 0
        for cc, cl, mask, back, e in ents:
            if cc != c:
                continue
            assert mask & back == 0 and (mask | back) & seen == 0
            seen |= mask | back
        assert seen == (0xFFFFFFFF << (32 - cnt)) & 0xFFFFFFFF, (c, hex(seen), cnt)
    # records of one mask share their plane bit for bit (up to the sign of a zero); back masks hold the reversed normal; axis entries are axis planes
    for c, cl, mask, back, e in ents:
        for j in range(32):
            for m, sg in ((mask, 1.0), (back, -1.0)):
                if not (m >> (31 - j)) & 1:
                    continue
                r = prep[32 * c + j]
                nrm = np.array([r[3], r[7], r[11]], np.float32) * np.float32(sg) + np.float32(0.0)
                if cl < 3:
                    assert r[cl] + np.float32(0) == e.view(np.float32)[0] and nrm[cl] == e.view(np.float32)[1]
                    assert nrm[(cl + 1) % 3] == 0 and nrm[(cl + 2) % 3] == 0 and r[4 + cl] == 0 and r[8 + cl] == 0
                else:
                    assert np.array_equal(nrm, e.view(np.float32)[:3] + np.float32(0))
    if not shuffle and n >= 12:
        assert len(ents) < n          # the halves of the quads did merge


@pytest.mark.parametrize("n", [2, 12, 33, 64, 96])
def test_sweep_words_keep_every_record_the_reference_accepts(n):
    """Rays aimed at a point of record i from its front side, windows around the hit: bit 31 - (i % 32) of word i // 32 -- the bit the candidate
    loop maps back to record i -- must be set whenever the reference accepts record i, for EVERY record the reference accepts with that ray."""
    lib = _lib()
    rng = np.random.default_rng(n)
    prep = soup(rng, n, True)
    words = np.zeros(16 + 4 * 320, np.uint32)
    lib.oracle_plane_list(prep.ctypes.data, n, words.ctypes.data)
    accepted = kept = 0
    for it in range(3000):
        i = int(rng.integers(0, n))
        r = prep[i]
        p0, e1, e2, nn = r[0:3], r[4:7], r[8:11], np.array([r[3], r[7], r[11]])
        b, g = rng.random(), rng.random()
        if b + g > 1:
            b, g = 1 - b, 1 - g
        P = p0 + b * e1 + g * e2
        o = (P - nn / (np.linalg.norm(nn) + 1e-30) * rng.uniform(0.2, 2.0) + rng.normal(size=3) * 0.2).astype(np.float32)
        d = (P - o).astype(np.float32)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        t = float(np.linalg.norm(P - o))
        cmin, cmax, maxt = (0.0, t * 1.5, np.inf) if it % 3 else (t * 0.5, t * 4.0, t * 2.0)
        cand = np.zeros(3, np.uint32)
        lib.oracle_sweep_words(words.ctypes.data, n, o.ctypes.data, d.ctypes.data, cmin, cmax, maxt, cand.ctypes.data)
        for j in range(n):
            bit = (int(cand[j // 32]) >> (31 - j % 32)) & 1
            kept += bit
            if lib.oracle_sweep_ref_accepts(prep.ctypes.data, j, o.ctypes.data, d.ctypes.data, cmin, cmax, maxt):
                accepted += 1
                assert bit, (it, i, j)
    assert accepted > 1500                      # the aimed-at record is usually hit ...
    assert n < 12 or kept < 0.5 * 3000 * n      # ... and the words are a filter (one quad alone is one plane: both halves stay)
