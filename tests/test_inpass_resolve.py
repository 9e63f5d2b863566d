"""GPU: copyToPixel INSIDE the fused pass (A10 code.cl:1366-1386 folded into mirt_render_first_pass; pt_kernels_fused.hip resolve_block).

A frame's first pass at a ray count that puts whole pixels into a block of 256 ray ids (rays_per_pixel divides 256) sums each pixel's
accumulators in the block's LDS, in the reference's order, and writes `pixel` / `radiance` itself; `acu` may then be NULL.  Everything is
compared with the compiled reference's fixtures, tolerance 0: the frame, the radiance sums, the seeds -- and the per-ray accumulators
when the caller kept them.  The unit of the optimistic / exact kernel pair becomes the block (a block with a sample outside the guard
windows is re-run whole by the exact kernel): covered by the scene that defers (own_flat) and by the exact-only mode."""
import numpy as np
import pytest

import a10_pass as A
from conftest import FULL_CASES, bits, load_fixture

pytestmark = pytest.mark.gpu

E_ARG = -1   # MIRT_E_ARG (include/mirt.h)
RESOLVABLE = [n for n in FULL_CASES if 256 % int(n.rsplit("_r", 1)[1]) == 0]


@pytest.fixture(scope="module")
def ctx(pkg):
    from raytracing_amd.pyhost import mirt
    c = mirt.Context(0)
    yield c
    c.destroy()


@pytest.fixture(scope="module")
def ctx_sep(pkg):
    """a context that never resolves in the pass (MIRT_INPASS_RESOLVE=0 is read when a context is created): the separate copyToPixel, as the
    baseline the in-pass sums are compared with -- since later passes resolve in the pass too, a non-first pass no longer is that baseline"""
    import os
    from raytracing_amd.pyhost import mirt
    os.environ["MIRT_INPASS_RESOLVE"] = "0"
    try:
        c = mirt.Context(0)
    finally:
        del os.environ["MIRT_INPASS_RESOLVE"]
    yield c
    c.destroy()


def check_frame(fr, fx, tag, rows=None):
    pix, rad = fx["pixel"], fx["radiance"]
    if rows is not None:
        lo, hi = rows
        pix, rad = pix[lo:hi], rad[lo:hi]
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), pix), tag + ": pixel"
    assert np.array_equal(bits(fr.radiance.read(np.float32).reshape(-1, 4)), bits(rad)), tag + ": radiance"


@pytest.mark.parametrize("exact_only", [False, True], ids=["optimistic", "exact_only"])
@pytest.mark.parametrize("keep_acu", [True, False], ids=["acu", "no_acu"])
@pytest.mark.parametrize("name", RESOLVABLE)
def test_first_pass_resolves_its_pixels(ctx, pkg, name, keep_acu, exact_only):
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture(name)
    ctx.set_exact_only(exact_only)
    try:
        fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"], keep_acu=keep_acu)
        if keep_acu:   # poison: the first pass must not read it, and must overwrite all of it
            fr.acu.write(np.full(sc.total_rays * 4, np.nan, np.float32))
        fr.pixel.write(np.full(sc.width * sc.height * 4, 7, np.uint8))
        fr.execute_render(fresh=True)
        check_frame(fr, fx, name)
        assert np.array_equal(fr.seeds.read(np.int32), fx["f_seeds"]), "seeds"
        if keep_acu:
            assert np.array_equal(bits(fr.acu.read(np.float32).reshape(-1, 4)), bits(fx["f_acu"])), "acu"
        fr.release()
    finally:
        ctx.set_exact_only(False)


def test_deferred_blocks_are_rerun_whole(ctx, ctx_sep, pkg):
    """own_flat: axis-parallel geometry, 0.6 % of its samples leave the optimistic kernel's guard windows.  With the pass resolving its
    pixels the exact kernel re-runs every BLOCK that holds one: mirt_pass_deferred counts whole blocks, and the frame is the fixture's."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("own_flat_32x24_r4")
    fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"], keep_acu=False)
    fr.execute_render(fresh=True)
    d_blocks = ctx.pass_deferred()
    check_frame(fr, fx, "own_flat")
    assert np.array_equal(fr.seeds.read(np.int32), fx["f_seeds"])
    fr.release()
    fr = render.FusedRenderer(ctx_sep, sc, seeds=fx["seeds_in"])
    fr.execute_render(fresh=False)            # the separate copyToPixel: a bit per sample
    d_samples = ctx_sep.pass_deferred()
    check_frame(fr, fx, "own_flat, separate copyToPixel")
    fr.release()
    assert d_samples > 0 and d_samples % 256 != 0 and d_blocks % 256 == 0 and d_samples <= d_blocks <= 256 * d_samples
    fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"])
    fr.execute_render(fresh=False)            # a later pass (the accumulator read and written) resolves in the pass as well: whole blocks again
    assert ctx.pass_deferred() == d_blocks
    check_frame(fr, fx, "own_flat, non-first pass")
    fr.release()


@pytest.mark.parametrize("name", ["cornell_teapot3_32x24_r4", "cornell_32x24_r4", "own_gems_48x36_r4"])
def test_row_tiles_with_partial_blocks(ctx, pkg, name):
    """Tiles whose ray count is not a multiple of 256: the last block holds pixels that do not exist (and, in the grid kernels, lanes that
    ride along on the tile's last sample): nothing of them is written, the tile's own pixels are the fixture's."""
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture(name)
    for row0, nrows in [(3, 7), (0, 1), (sc.height - 5, 5)]:
        fr = render.FusedRenderer(ctx, sc, row0=row0, nrows=nrows, keep_acu=False)
        guard = np.full(nrows * sc.width * 4 + 64, 0xAB, np.uint8)
        fr.pixel.release()
        fr.pixel = ctx.buffer(guard.size)
        fr.pixel.write(guard)
        fr.execute_render(fresh=True)
        got = fr.pixel.read(np.uint8)
        assert np.array_equal(got[:nrows * sc.width * 4].reshape(-1, 4), fx["pixel"][row0 * sc.width:(row0 + nrows) * sc.width]), (name, row0, nrows)
        assert np.all(got[nrows * sc.width * 4:] == 0xAB), "wrote past the tile's last pixel"
        assert np.array_equal(bits(fr.radiance.read(np.float32).reshape(-1, 4)), bits(fx["radiance"][row0 * sc.width:(row0 + nrows) * sc.width]))
        fr.release()


def test_resolve_equals_the_separate_kernel_at_depth_8_and_larger_counts(ctx, ctx_sep, pkg):
    """rays_per_pixel 16, 64 and 256 (one, four and sixteen... pixels per block down to one): the in-pass sums against the separate
    copyToPixel of the same library (a context created with MIRT_INPASS_RESOLVE=0), and against the CPU oracle's frame."""
    from raytracing_amd.pyhost import render, scene
    fx, sc0 = load_fixture("cornell_teapot3_32x24_r4")
    for rpp, (w, h) in ((16, (40, 30)), (64, (24, 18)), (256, (12, 9))):
        ps = scene.PackedScene(dict(sc0.d)).resized(w, h, rpp)
        sc = A.Scene(ps.d)
        seeds = A.make_seeds(sc.total_rays, seed_base=rpp)
        a = render.FusedRenderer(ctx, ps, seeds=seeds, keep_acu=False)
        a.execute_render(bounces=8, fresh=True)
        b = render.FusedRenderer(ctx_sep, ps, seeds=seeds)
        b.execute_render(bounces=8, fresh=False)
        assert np.array_equal(a.pixel.read(np.uint8), b.pixel.read(np.uint8)), rpp
        assert np.array_equal(bits(a.radiance.read(np.float32)), bits(b.radiance.read(np.float32))), rpp
        assert np.array_equal(a.seeds.read(np.int32), b.seeds.read(np.int32)), rpp
        st = A.PassState(sc, seeds)
        A.run_pass(A.load_oracle(), sc, st, bounces=8)
        assert np.array_equal(a.pixel.read(np.uint8).reshape(-1, 4), st.pixel), rpp
        a.release()
        b.release()


def test_acu_is_only_optional_where_the_pass_resolves(ctx, pkg):
    from raytracing_amd.pyhost import mirt, render
    fx, sc = load_fixture("cornell_16x12_r9")          # nine rays per pixel: pixels straddle blocks
    fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"], keep_acu=False)
    with pytest.raises(mirt.MirtError) as e:
        fr.execute_render(fresh=True)
    assert e.value.code == E_ARG and "acu" in str(e.value)
    fr.release()
    fx, sc = load_fixture("cornell_32x24_r4")
    fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"], keep_acu=False)
    with pytest.raises(mirt.MirtError) as e:
        fr.execute_render(fresh=False)                  # not a first pass: there is an accumulator to read
    assert e.value.code == E_ARG
    d = fr.dev.pass_desc(fr.seeds, None, None, None)    # nowhere to put the frame
    with pytest.raises(mirt.MirtError) as e:
        ctx.render_pass(d, fresh=True)
    assert e.value.code == E_ARG
    fr.execute_render(fresh=True)                       # and the context is fine afterwards
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), fx["pixel"])
    fr.release()


@pytest.mark.parametrize("exact_only", [False, True], ids=["optimistic", "exact_only"])
@pytest.mark.parametrize("rpp,size", [(1024, (7, 5)), (4096, (3, 2))])
@pytest.mark.parametrize("name", ["cornell_32x24_r4", "cornell_teapot3_32x24_r4", "own_flat_32x24_r4"])
def test_pixels_of_more_than_256_rays_resolve_block_by_block(ctx, ctx_sep, pkg, name, rpp, size, exact_only):
    """BASELINE config 5's ray count (1024 = a 32 x 32 lens grid: four blocks of 256 ray ids per pixel) and the next square that is 256 times a
    power of two (4096: sixteen): the pass runs once per block of a pixel, in ray order, each launch going on from the sums the one before left
    (pt_launch.hpp FusedArgs::chunks) -- the reference's single chain of additions (A10 code.cl:1377-1380), cut at multiples of 256.  Against the
    separate copyToPixel of the same library over a kept accumulator (bit for bit: frame, radiance, seeds), with and without a radiance buffer of
    the caller's, on a scene with grid meshes, on one that defers blocks to the exact kernel (own_flat), and -- 1024 rays -- against the CPU oracle."""
    from raytracing_amd.pyhost import render, scene
    fx, sc0 = load_fixture(name)
    w, h = size
    d = dict(sc0.d)
    ps = scene.PackedScene(d).resized(w, h, rpp)
    sc = A.Scene(ps.d)
    seeds = A.make_seeds(sc.total_rays, seed_base=rpp + w)
    ctx.set_exact_only(exact_only)
    ctx_sep.set_exact_only(exact_only)
    try:
        b = render.FusedRenderer(ctx_sep, ps, seeds=seeds)
        b.execute_render(fresh=False)                                   # accumulator + the separate copyToPixel
        deferred_samples = ctx_sep.pass_deferred()
        for want_radiance in (True, False):
            a = render.FusedRenderer(ctx, ps, seeds=seeds, keep_acu=False, want_radiance=want_radiance)
            a.pixel.write(np.full(sc.width * sc.height * 4, 7, np.uint8))
            a.execute_render(fresh=True)
            deferred_blocks = ctx.pass_deferred()
            assert np.array_equal(a.pixel.read(np.uint8), b.pixel.read(np.uint8)), (rpp, want_radiance)
            if want_radiance:
                assert np.array_equal(bits(a.radiance.read(np.float32)), bits(b.radiance.read(np.float32))), rpp
            assert np.array_equal(a.seeds.read(np.int32), b.seeds.read(np.int32)), rpp
            assert deferred_blocks % 256 == 0 and deferred_samples <= deferred_blocks <= 256 * deferred_samples
            a.release()
        # and with the accumulator kept beside the in-pass resolve: every per-ray value too
        a = render.FusedRenderer(ctx, ps, seeds=seeds, keep_acu=True)
        a.acu.write(np.full(sc.total_rays * 4, np.nan, np.float32))
        a.execute_render(fresh=True)
        assert np.array_equal(bits(a.acu.read(np.float32)), bits(b.acu.read(np.float32)))
        assert np.array_equal(a.pixel.read(np.uint8), b.pixel.read(np.uint8))
        a.release()
        if name == "own_flat_32x24_r4" and not exact_only:
            assert deferred_samples > 0, "own_flat no longer defers: the redo launches between the chunks are not exercised"
        if rpp == 1024 and not exact_only:
            st = A.PassState(sc, seeds)
            A.run_pass(A.load_oracle(), sc, st)
            assert np.array_equal(b.pixel.read(np.uint8).reshape(-1, 4), st.pixel)
        # a later pass of a progressive frame resolves block by block too: two passes against two passes of the separate kernel
        a = render.FusedRenderer(ctx, ps, seeds=seeds)
        for _ in range(2):
            a.execute_render(fresh=False)
        b.execute_render(fresh=False)
        assert np.array_equal(a.pixel.read(np.uint8), b.pixel.read(np.uint8)) and np.array_equal(bits(a.acu.read(np.float32)), bits(b.acu.read(np.float32)))
        assert np.array_equal(bits(a.radiance.read(np.float32)), bits(b.radiance.read(np.float32)))
        a.release()
        b.release()
    finally:
        ctx.set_exact_only(False)
        ctx_sep.set_exact_only(False)


def test_row_tiles_of_pixels_of_1024_rays(ctx, ctx_sep, pkg):
    """a row tile of the block-by-block resolve: its pixels are the whole frame's, nothing is written past them"""
    from raytracing_amd.pyhost import render, scene
    fx, sc0 = load_fixture("cornell_teapot3_32x24_r4")
    ps = scene.PackedScene(dict(sc0.d)).resized(5, 6, 1024)
    sc = A.Scene(ps.d)
    seeds = A.make_seeds(sc.total_rays, seed_base=99)
    whole = render.FusedRenderer(ctx_sep, ps, seeds=seeds)
    whole.execute_render(fresh=False)
    want = whole.pixel.read(np.uint8).reshape(-1, 4)
    for row0, nrows in [(0, 1), (2, 3), (5, 1)]:
        fr = render.FusedRenderer(ctx, ps, seeds=seeds, row0=row0, nrows=nrows, keep_acu=False)
        guard = np.full(nrows * 5 * 4 + 64, 0xAB, np.uint8)
        fr.pixel.release()
        fr.pixel = ctx.buffer(guard.size)
        fr.pixel.write(guard)
        fr.execute_render(fresh=True)
        got = fr.pixel.read(np.uint8)
        assert np.array_equal(got[:nrows * 5 * 4].reshape(-1, 4), want[row0 * 5:(row0 + nrows) * 5]), (row0, nrows)
        assert np.all(got[nrows * 5 * 4:] == 0xAB)
        fr.release()
    whole.release()
