"""C ABI robustness (include/mirt.h "Conventions"): typed handle validation, contexts reaping what the host left behind,
recordings pinned to the allocations they captured, 64-bit slot counting in the grid builder, device groups + gather."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_ARG, E_HANDLE, E_RANGE = -1, -2, -5


@pytest.fixture()
def M(pkg):
    from raytracing_amd.pyhost import mirt
    return mirt


def code(fn, *a, **kw):
    with pytest.raises(Exception) as e:
        fn(*a, **kw)
    return e.value.code


def test_handles_are_validated_by_kind(M):
    ctx = M.Context(0)
    buf = ctx.buffer(64)
    k = C.c_void_p()
    assert M.lib().mirt_kernel_get(ctx.h, b"initAcu", C.byref(k)) == 0
    # a kernel handle where a buffer is expected, a buffer where a kernel is expected, a context as a buffer
    assert M.lib().mirt_kernel_set_arg_buf(k, 0, k) == E_HANDLE
    assert M.lib().mirt_kernel_set_arg_buf(buf.h, 0, buf.h) == E_HANDLE
    assert M.lib().mirt_buf_release(ctx.h) == E_HANDLE
    assert M.lib().mirt_buf_size(k) == 0
    assert M.lib().mirt_kernel_release(k) == 0 and M.lib().mirt_kernel_release(k) == E_HANDLE   # double release
    buf.release()
    ctx.destroy()


def test_context_reaps_what_the_host_left_behind(M):
    """The reference host never releases its bouncePaths kernel (A10 code.js:1444-1455: not pushed on cl_resources) and then releases
    the context: the runtime frees the leftovers with the context and their handles turn invalid -- no leak, no dangling context."""
    ctx = M.Context(0)
    buf = ctx.buffer(1 << 20)
    k = C.c_void_p()
    assert M.lib().mirt_kernel_get(ctx.h, b"bouncePaths", C.byref(k)) == 0
    ctx.destroy()
    assert M.lib().mirt_kernel_set_arg(k, 3, 4, C.byref(C.c_uint32(1))) == E_HANDLE
    assert M.lib().mirt_buf_release(buf.h) == E_HANDLE
    assert M.lib().mirt_kernel_release(k) == E_HANDLE


def test_a_recording_is_refused_once_its_allocations_are_gone(M):
    ctx = M.Context(0)
    n = 4096
    acu, acu2 = ctx.buffer(n * 16), ctx.buffer(n * 16)
    k = C.c_void_p()
    assert M.lib().mirt_kernel_get(ctx.h, b"initAcu", C.byref(k)) == 0
    M.lib().mirt_kernel_set_arg_buf(k, 0, acu.h)
    M.lib().mirt_kernel_set_arg(k, 1, 4, C.byref(C.c_uint32(n)))
    g1 = (C.c_size_t * 1)(n)
    assert M.lib().mirt_enqueue(ctx.h, k, 1, g1, None) == 0
    ctx.finish()
    ctx.capture_begin()
    assert M.lib().mirt_enqueue(ctx.h, k, 1, g1, None) == 0
    graph = ctx.capture_end()
    ctx.graph_launch(graph)                       # fine while everything it captured is alive
    ctx.finish()
    acu2.release()                                # an unrelated buffer: the recording is untouched
    ctx.graph_launch(graph)
    ctx.finish()
    acu.release()                                 # the captured allocation: replay would write into freed memory
    assert code(ctx.graph_launch, graph) == E_HANDLE
    assert "record the sequence again" in ctx.last_error()
    again = ctx.buffer(n * 16)                    # even if the allocator hands the same address out again
    assert code(ctx.graph_launch, graph) == E_HANDLE
    again.release()
    ctx.destroy()                                 # reaps the kernel and the graph


def test_a_recording_is_refused_once_its_geometry_is_rewritten(M, pkg):
    """Cell-offset tables are validated and triangles prepared on the host side of a launch; a recording replays neither, so it pins the
    contents it was validated against."""
    import a10_pass as A
    from conftest import load_fixture
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("own_gems_48x36_r4")
    ctx = M.Context(0)
    gr = render.GranularRenderer(ctx, sc, seeds=A.make_seeds(sc.total_rays))
    gr.execute_render()
    gr.execute_render(use_graph=True)             # records the pass body
    gr.execute_render(use_graph=True)             # replays it
    graph = gr._graph
    assert graph is not None
    tri = gr.dev.meshes[0]["prims"]
    data = tri.read(np.float32)
    tri.write(data)                               # same bytes, new version: the prepared copy is no longer known to match
    assert code(ctx.graph_launch, graph) == E_HANDLE
    gr.release()
    ctx.destroy()


def test_grid_build_counts_slots_in_64_bits(M):
    """Five primitives that cover every cell of a 1024^3 grid: 5 * 2^30 slots do not fit 32 bits.  Counted in 64 bits and refused."""
    ctx = M.Context(0)
    tri = np.tile(np.array([-9, -9, -9, 9, 9, -9, 9, -9, 9], np.float64), (5, 1))
    assert code(ctx.grid_build, 1, tri, [-1, -1, -1, 1, 1, 1], 1024) == E_RANGE
    assert "5368709120" in ctx.last_error()
    off, order, total = ctx.grid_build(1, tri, [-1, -1, -1, 1, 1, 1], 8)     # the same primitives on a grid that fits
    assert total == 5 * 512 and off.read(np.uint32)[-1] == total
    off.release(); order.release()
    ctx.destroy()


@pytest.mark.parametrize("use_rccl", [False, True])
def test_device_group_renders_row_tiles_and_gathers(M, pkg, use_rccl):
    """mirt_group over the one device this box has, one context: its tile is the frame; gathered by the device copy and through RCCL with a
    one-rank communicator (ncclSend / ncclRecv to self inside one group) == the compiled reference's frame.  N > 1: the next test."""
    import a10_pass as A
    from conftest import load_fixture
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("cornell_32x24_r4")
    grp = M.DeviceGroup([0])
    ctx = grp.contexts[0]
    assert M.lib().mirt_ctx_destroy(ctx.h) == E_ARG       # a group's context goes with the group
    seeds = A.make_seeds(sc.total_rays)
    assert grp.tile_rows(sc.height, 0) == (0, sc.height)
    fr = render.FusedRenderer(ctx, sc, seeds=seeds)
    fr.execute_render()
    frame = ctx.buffer(sc.width * sc.height * 4)
    grp.gather([fr.pixel], [sc.width * sc.height * 4], frame, root=0, use_rccl=use_rccl)
    assert grp.routes() == ["rccl" if use_rccl else "local"]   # what the gather did, as the group reports it (mirt_gather_route)
    grp.finish()
    assert np.array_equal(frame.read(np.uint8).reshape(-1, 4), fx["pixel"])
    # tile arithmetic: contiguous, sizes differ by at most one row, cover the image
    for h, n in ((1080, 8), (1081, 8), (7, 8), (2160, 3)):
        r0 = C.c_uint32(); nr = C.c_uint32(); acc = 0
        for i in range(n):
            M.lib().mirt_tile_rows(h, n, i, C.byref(r0), C.byref(nr))
            assert r0.value == acc and nr.value in (h // n, h // n + 1)
            acc += nr.value
        assert acc == h
    fr.release()
    grp.destroy()
    assert M.lib().mirt_finish(ctx.h) == E_HANDLE      # the group took its contexts with it


@pytest.mark.parametrize("n", [2, 3, 5, 8])
def test_device_group_n_contexts_render_tiles_and_gather(M, pkg, n, monkeypatch):
    """The C-ABI N-device path with N > 1 on the one GPU reachable here (the rehearsal switch of include/mirt.h): N contexts on device 0,
    each renders ITS mirt_tile_rows tile of cornell_teapot3 (grid meshes, two lights; 24 rows over 5 tiles = uneven heights) with global
    ray ids, mirt_gather assembles the RGBA8 and the radiance tiles at their offsets by copies ordered after each context's own stream:
    frame and radiance == the compiled reference's.  Everything of the N > 1 path except the transport over xGMI."""
    import a10_pass as A
    from conftest import load_fixture
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("cornell_teapot3_32x24_r4")
    assert code(M.DeviceGroup, [0] * n) == E_ARG            # a device listed twice is refused ...
    monkeypatch.setenv("MIRT_GROUP_ALLOW_REPEATED_DEVICES", "1")   # ... unless the rehearsal switch is on
    grp = M.DeviceGroup([0] * n)
    seeds = fx["seeds_in"]
    frs, acc = [], 0
    for i, ctx in enumerate(grp.contexts):
        row0, nrows = grp.tile_rows(sc.height, i)
        assert row0 == acc and nrows in (sc.height // n, sc.height // n + 1)
        acc += nrows
        fr = render.FusedRenderer(ctx, sc, seeds=seeds, row0=row0, nrows=nrows, want_radiance=True)
        fr.execute_render()                                  # asynchronous: the N passes are queued on N streams
        frs.append(fr)
    assert acc == sc.height
    root = grp.contexts[0]
    npix = sc.width * sc.height
    frame, rad = root.buffer(npix * 4), root.buffer(npix * 16)
    assert code(grp.gather, [f.pixel for f in frs], [f.npix * 4 for f in frs], frame, transport=M.DeviceGroup.GATHER_RCCL) == E_ARG   # one device per rank
    assert code(grp.gather, [f.pixel for f in frs][:-1], [f.npix * 4 for f in frs][:-1], frame) == E_ARG                               # one tile per context
    assert grp.routes() == ["none"] * n                      # nothing gathered yet
    assert all(grp.peer_access(i, j) == 1 for i in range(n) for j in range(n))   # contexts that share a device reach each other's memory
    assert grp.peer_access(0, n) < 0 and grp.peer_access(-1, 0) < 0
    grp.gather([f.pixel for f in frs], [f.npix * 4 for f in frs], frame)
    assert grp.routes() == ["local"] * n                     # AUTO on a rehearsal group: device-local copies, and the group says so
    grp.gather([f.radiance for f in frs], [f.npix * 16 for f in frs], rad, transport=M.DeviceGroup.GATHER_COPY)
    assert grp.routes() == ["local"] * n
    grp.finish()
    assert np.array_equal(frame.read(np.uint8).reshape(-1, 4), fx["pixel"])
    assert np.array_equal(rad.read(np.float32).view(np.uint32).reshape(-1, 4), fx["radiance"].view(np.uint32).reshape(-1, 4))
    for f in frs:
        f.release()
    grp.destroy()


def test_two_sets_of_a_pass_may_not_share_a_position_buffer_with_different_counts(M, pkg):
    """The runtime keeps ONE prepared copy per position buffer, laid out for one record count (records, group spheres, the sweep's plane list):
    a pass whose loose-triangle set and a mesh share a buffer but hold different numbers of slots would have the second set re-lay the copy the
    first set's pointers describe.  Refused, with both counts named; the same two sets on their own buffers render."""
    import a10_pass as A
    from conftest import load_fixture
    from raytracing_amd.pyhost import render
    fx, sc = load_fixture("cornell_teapot3_32x24_r4")
    ctx = M.Context(0)
    fr = render.FusedRenderer(ctx, sc, seeds=fx["seeds_in"])
    box = fr.dev.meshes[1]                                         # the 20-triangle box in 5^3 cells: 210 slots
    keep = dict(fr.dev.tri)
    fr.dev.tri = dict(fr.dev.tri, prims=box["prims"], normals=box["normals"])   # the loose triangles read out of the box's buffers
    with pytest.raises(M.MirtError) as e:
        fr.execute_render()
    n_loose, n_box = int(np.asarray(sc.t_box)[-1]), int(np.asarray(sc.meshes[1]["box"])[-1])
    assert n_loose != n_box and e.value.code == E_ARG and f"{n_loose} and {n_box} slots" in str(e.value)
    fr.dev.tri = keep
    fr.execute_render()
    assert np.array_equal(fr.pixel.read(np.uint8).reshape(-1, 4), fx["pixel"])
    fr.release()
    ctx.destroy()
