"""GPU: the numerics contract on gfx950 -- correctly rounded / and sqrt, the shared sin/cos, the
LCG, min/max forms -- compared bit for bit with the CPU model's primitives (oracle/cl_numerics.h
via liboracle.so probes, numpy for IEEE / and sqrt) over millions of inputs.  The comparison of every built-in with
AMD's OpenCL library itself is tests/test_ref_gpu.py."""
import ctypes as C

import numpy as np
import pytest

import a10_pass as A

pytestmark = pytest.mark.gpu

N = 1 << 21


@pytest.fixture(scope="module")
def ctx(pkg):
    from raytracing_amd.pyhost import mirt
    c = mirt.Context(0)
    yield c
    c.destroy()


def rnd(n, seed, lo=-1e3, hi=1e3):
    r = np.random.default_rng(seed)
    # mix magnitudes: uniform, tiny, huge, denormal-producing
    a = r.uniform(lo, hi, n).astype(np.float32)
    a[: n // 8] *= np.float32(1e-30)
    a[n // 8: n // 4] *= np.float32(1e30)
    a[n // 4: n // 4 + n // 16] = r.uniform(-1e-38, 1e-38, n // 16).astype(np.float32)
    return a


def test_division_is_correctly_rounded(ctx):
    a, b = rnd(N, 1), rnd(N, 2)
    got = ctx.debug_numerics(0, a, b)
    with np.errstate(all="ignore"):
        want = (a / b).astype(np.float32)
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert ok.all(), f"{(~ok).sum()} of {N} quotients differ from IEEE"


def test_sqrt_is_correctly_rounded(ctx):
    a = np.abs(rnd(N, 3))
    got = ctx.debug_numerics(1, a)
    want = np.sqrt(a).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_sincos_bits_match_cpu(ctx):
    lib = A.load_oracle().lib
    lib.oracle_bi_sin.restype = C.c_float
    lib.oracle_bi_sin.argtypes = [C.c_float]
    lib.oracle_bi_cos.restype = C.c_float
    lib.oracle_bi_cos.argtypes = [C.c_float]
    x = np.concatenate([np.linspace(-np.pi / 4, 3 * np.pi / 4, 200001), np.linspace(-20, 20, 50001)]).astype(np.float32)
    s_dev, c_dev = ctx.debug_numerics(2, x), ctx.debug_numerics(3, x)
    s_cpu = np.array([lib.oracle_bi_sin(float(v)) for v in x], np.float32)
    c_cpu = np.array([lib.oracle_bi_cos(float(v)) for v in x], np.float32)
    assert np.array_equal(s_dev.view(np.uint32), s_cpu.view(np.uint32))
    assert np.array_equal(c_dev.view(np.uint32), c_cpu.view(np.uint32))
    core = slice(0, 200001)
    assert np.abs(s_dev[core] - np.sin(x[core].astype(np.float64))).max() < 2.5e-7   # <= ~2 ulp near 1


def test_lcg_matches_reference_known_answers(ctx):
    # 42 -> 705894 -> -1020941430 -> -568266490 -> 1152368874 (compiled reference)
    s = np.array([42], np.int32)
    seq = []
    for _ in range(4):
        s = ctx.debug_numerics(5, s.view(np.float32)).view(np.int32)
        seq.append(int(s[0]))
    assert seq == [705894, -1020941430, -568266490, 1152368874]
    # random states incl. the three special residues of x % (2^31-1)
    r = np.random.default_rng(7).integers(-2**31, 2**31, N, dtype=np.int64).astype(np.int32)
    got = ctx.debug_numerics(5, r.view(np.float32)).view(np.int32)
    w = (r.astype(np.int64) * 16807).astype(np.int32).astype(np.int64)      # int32 wrap, then widen
    want = (np.sign(w) * (np.abs(w) % 2147483647)).astype(np.int32)          # C truncating remainder
    assert np.array_equal(got, want)
    f = ctx.debug_numerics(4, r.view(np.float32))
    assert np.array_equal(f.view(np.uint32), np.abs(want.astype(np.float32) * np.float32(2.0**-31)).view(np.uint32))


def test_min_max_forms(ctx):
    """min / max / fmin / fmax are all v_min_f32 / v_max_f32 (AMD's OpenCL library: llvm.minnum / maxnum): a NaN loses, -0 < +0."""
    a, b = rnd(1 << 16, 11), rnd(1 << 16, 12)
    a[:100] = np.nan
    b[50:150] = np.nan
    a[200:210], b[200:210] = 0.0, -0.0
    a[210:220], b[210:220] = -0.0, 0.0
    with np.errstate(all="ignore"):
        want_min, want_max = np.fmin(a, b), np.fmax(a, b)
    z = (a == 0) & (b == 0)
    want_min[z] = np.where(np.signbit(a[z]) | np.signbit(b[z]), np.float32(-0.0), np.float32(0.0))
    want_max[z] = np.where(np.signbit(a[z]) & np.signbit(b[z]), np.float32(-0.0), np.float32(0.0))
    from conftest import bits
    for op in (6, 8):
        assert np.array_equal(bits(ctx.debug_numerics(op, a, b)), bits(want_min)), op
    for op in (7, 9):
        assert np.array_equal(bits(ctx.debug_numerics(op, a, b)), bits(want_max)), op


def test_concentric_map_matches_cpu_formula(ctx):
    r = np.random.default_rng(5)
    u, v = r.random(1 << 18).astype(np.float32), r.random(1 << 18).astype(np.float32)
    x, y = ctx.debug_numerics(11, u, v), ctx.debug_numerics(12, u, v)
    assert (x * x + y * y <= 1.0 + 1e-6).all()
    # exact comparison against the CPU pipeline happens in test_gpu_parity (every ray goes through it)


def test_cheap_exact_division_forms(ctx):
    """The 3-operation reciprocal and shared-reciprocal quotient the hot loops use (pt_numerics.hpp) against the
    compiler's 11-operation correctly rounded division, on the device."""
    r = ctx.divcheck(3, 0, 1 << 32)
    # every bit pattern: the only denominators where rcp_refined != 1/d are +-0, +-inf, denormals and |d| > 2^126
    assert int(r[1]) == 3 * 2**24      # 2 zeros + 2 infs + 2(2^23 - 1) denormals + 2(2^24 - 1) patterns with |d| > 2^126
    for mode, count in ((0, 1 << 31), (1, 81 << 23), (2, 1 << 33)):
        r = ctx.divcheck(mode, 4242 + mode, count)
        assert int(r[12]) == 0, f"div_exact3 differs from n/d on {int(r[12])} of {count} pairs (mode {mode})"
        assert int(r[1]) == 0 and int(r[2]) == 0
        if mode == 2:
            assert int(r[0]) == 0 and int(r[3]) == 0       # without +-0 numerators even the bare forms agree
    # the 9-operation sqrt: every bit pattern; the bare core may only differ from IEEE inside (0, 2^-96), where cl_sqrt falls back
    r = ctx.divcheck(5, 0, 1 << 32)
    assert int(r[1]) == 0, f"cl_sqrt differs from the correctly rounded sqrt on {int(r[1])} bit patterns"
    assert int(r[3]) == 0 and int(r[2]) < 31 * 2**23
    # a slice of the exhaustive mantissa-pair sweep (all 2^46 pairs: profiles/divcheck.py --exhaustive, result under profiles/)
    for first in (0, 0x400000, 0x7FFF00):
        r = ctx.divcheck(4, first, 256)
        assert int(r[3]) == 0, f"div_exact3(1.nm, 1.dm) differs for dm near {first:#x}: n,d bits {int(r[6]):#x},{int(r[7]):#x}"
