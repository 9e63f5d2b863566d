"""Pass drivers over the C ABI, in Python, for tests and bench.py.

`GranularRenderer` is a line-by-line functional mirror of the reference host's kernel
plumbing -- preRender() / executeRender() of A10 code.js:1784-1854 with its prepare*/execute*
helpers (:1078-1528) -- written against pyhost.mirt instead of WebCL: same kernels, same
argument indices, same enqueue order, same NDRange padding.  It exists so that the parity
tests read like the reference's own call sequence.  `FusedRenderer` is the one-launch path
(mirt_render_pass).  The JavaScript host in ../host/ has the same two drivers; that is the
product, this is tooling.
"""
import math

import numpy as np

from . import mirt

RAY_BYTES, POI_BYTES = 48, 64


def _u32(v):
    return np.array([v], np.uint32)


def _f32(v):
    return np.array([v], np.float32)


def get_local_ws(dim, multiple):
    """getLocalWS (A10 code.js:645-672)."""
    if dim == 1:
        return [multiple]
    x = int(math.floor(math.sqrt(multiple)))
    if x & (x - 1):
        x = 1 << (x - 1).bit_length()
    return [x, multiple // x]


class GranularRenderer:
    def __init__(self, ctx, scene, seeds=None, seed_base=0):
        self.ctx, self.s = ctx, scene
        self.dev = mirt.DeviceScene(ctx, scene)
        self.k, self.b, self.gws, self.lws = {}, {}, {}, {}
        self.passes = 1
        self._pre_render(seeds, seed_base)

    # -- preRender (code.js:1784-1804)
    def _pre_render(self, seeds, seed_base):
        ctx, s = self.ctx, self.s
        n = s.total_rays
        self.total_rays = n
        # getStructSize (code.js:1064-1076)
        sizes = {}
        for name in ("Ray", "Poi"):
            k = ctx.kernel("sizeof" + name)
            tmp = ctx.buffer(4, mirt.MEM_WRITE_ONLY)
            k.set_arg(0, tmp)
            k.enqueue([1], [1])
            sizes[name] = int(tmp.read(np.uint32, 1)[0])
            tmp.release()
            k.release()
        self.ray_size, self.poi_size = sizes["Ray"], sizes["Poi"]

        # prepareInitAcu (code.js:1078-1099)
        self.b["acu"] = ctx.buffer(n * 16)
        k = ctx.kernel("initAcu").set_args(self.b["acu"], _u32(n))
        l = get_local_ws(1, 64)
        k.enqueue([-(-n // l[0]) * l[0]], l)
        ctx.finish()
        k.release()
        # prepareInitSeeds (code.js:1140-1154)
        self.b["seeds"] = ctx.buffer(n * 4)
        if seeds is not None:
            self.b["seeds"].write(np.asarray(seeds, np.int32))
        else:
            ctx.seed_fill(self.b["seeds"], 0, n, seed_base)
        # prepareInitTrace (code.js:1101-1138)
        self.b["rays"] = ctx.buffer(n * self.ray_size)
        self.b["pois"] = ctx.buffer(n * self.poi_size)
        self.k["initTrace"] = ctx.kernel("initTrace").set_args(
            self.b["seeds"], self.b["rays"], self.b["pois"], s.bounds, None, _f32(s.focal_length), _f32(s.lens_rad), _u32(s.rpp))
        l = get_local_ws(2, 64)
        self.lws["initTrace"] = l
        self.gws["initTrace"] = [-(-s.width // l[0]) * l[0], -(-s.height // l[1]) * l[1]]
        g1 = [-(-n // 64) * 64]
        d = self.dev
        if s.has_spheres:  # prepareSphereTrace (code.js:1156-1202)
            self.k["sphereTrace"] = ctx.kernel("sphereTrace").set_args(
                _u32(n), self.b["pois"], self.b["rays"], d.sph["prims"], d.sph["matid"], d.sph["off"], s.sphere_bounds, _u32(s.n_slabs))
        if s.has_triangles:  # prepareTriangleTrace (code.js:1204-1252)
            self.k["triangleTrace"] = ctx.kernel("triangleTrace").set_args(
                _u32(n), self.b["pois"], self.b["rays"], d.tri["prims"], d.tri["normals"], d.tri["matid"], d.tri["off"],
                s.triangle_bounds, _u32(s.n_slabs))
        if s.meshes:  # prepareMeshTrace (code.js:1254-1291)
            self.k["meshTrace"] = ctx.kernel("meshTrace").set_args(_u32(n), self.b["pois"], self.b["rays"])
        # prepareInitShadowTrace (code.js:1417-1442)
        self.b["shadow"] = ctx.buffer(n * self.ray_size)
        self.k["initShadowTrace"] = ctx.kernel("initShadowTrace").set_args(self.b["shadow"], self.b["pois"], _u32(n), None, self.b["seeds"])
        if s.has_spheres:  # prepareSphereShadowTrace (code.js:1461-1479)
            self.k["sphereShadowTrace"] = ctx.kernel("sphereShadowTrace").set_args(
                _u32(n), self.b["shadow"], d.sph["prims"], d.sph["off"], s.sphere_bounds, _u32(s.n_slabs))
        if s.has_triangles or s.meshes:  # prepareTriangleShadowTrace (code.js:1481-1499)
            self.k["triangleShadowTrace"] = ctx.kernel("triangleShadowTrace").set_args(_u32(n), self.b["shadow"])
        # prepareSceneRender (code.js:1364-1395)
        self.k["sceneRender"] = ctx.kernel("sceneRender").set_args(self.b["acu"], self.b["pois"], self.b["shadow"], d.material, None, _u32(n))
        # prepareCopyToPixel (code.js:1305-1326)
        npix = s.width * s.height
        self.b["pixel"] = ctx.buffer(npix * 4, mirt.MEM_WRITE_ONLY)
        self.k["copyToPixel"] = ctx.kernel("copyToPixel").set_args(self.b["pixel"], self.b["acu"], None, _u32(npix), _u32(s.rpp))
        self.gws["copyToPixel"] = [-(-npix // 64) * 64]
        # prepareBouncePaths (code.js:1444-1459), prepareLightRender (code.js:1346-1362)
        self.k["bouncePaths"] = ctx.kernel("bouncePaths").set_args(self.b["pois"], self.b["rays"], self.b["seeds"], _u32(n))
        self.k["lightRender"] = ctx.kernel("lightRender").set_args(self.b["pois"], self.b["rays"], self.b["acu"], None, _u32(n))
        self.g1 = g1

    def _closest(self):
        s, k, d = self.s, self.k, self.dev
        if s.has_spheres:
            k["sphereTrace"].enqueue(self.g1, [64])
        if s.has_triangles:
            k["triangleTrace"].enqueue(self.g1, [64])
        for m in d.meshes:  # executeMeshTrace (code.js:1293-1303)
            k["meshTrace"].set_arg(3, m["prims"]).set_arg(4, m["normals"]).set_arg(5, m["off"]).set_arg(6, _u32(m["matid"]))
            k["meshTrace"].set_arg(7, m["bounds"]).set_arg(8, _u32(m["n"]))
            k["meshTrace"].enqueue(self.g1, [64])

    def _direct(self):
        s, k, d = self.s, self.k, self.dev
        for l in s.lights:
            k["initShadowTrace"].set_arg(3, l["shadow"]).enqueue(self.g1, [64])
            if s.has_spheres:
                k["sphereShadowTrace"].enqueue(self.g1, [64])
            if s.has_triangles:  # executeTriangleShadowTrace (code.js:1514-1520)
                k["triangleShadowTrace"].set_arg(2, d.tri["prims"]).set_arg(3, d.tri["off"]).set_arg(4, s.triangle_bounds).set_arg(5, _u32(s.n_slabs))
                k["triangleShadowTrace"].enqueue(self.g1, [64])
            for m in d.meshes:  # executeMeshShadowTrace (code.js:1522-1528)
                k["triangleShadowTrace"].set_arg(2, m["prims"]).set_arg(3, m["off"]).set_arg(4, m["bounds"]).set_arg(5, _u32(m["n"]))
                k["triangleShadowTrace"].enqueue(self.g1, [64])
            k["sceneRender"].set_arg(4, l["scene"]).enqueue(self.g1, [64])  # executeSceneRender (code.js:1402-1408)
            if not getattr(self, "_recording", False):   # the reference finishes the queue here (code.js:1406); a recording cannot wait
                self.ctx.finish()

    # -- executeRender (code.js:1806-1854)
    def _enqueue_segments(self, bounces, on_primary=None):
        s, k = self.s, self.k
        k["initTrace"].set_arg(4, s.cam).enqueue(self.gws["initTrace"], self.lws["initTrace"])
        self._closest()
        for l in s.lights:
            k["lightRender"].set_arg(3, l["light"]).enqueue(self.g1, [64])
        self._direct()
        if on_primary:
            on_primary(self)
        for _ in range(bounces):
            k["bouncePaths"].enqueue(self.g1, [64])
            self._closest()
            self._direct()

    def execute_render(self, bounces=5, on_primary=None, use_graph=False):
        """One executeRender() (code.js:1806-1854).  use_graph: the 40-odd enqueues of the pass body are recorded once (on the second
        pass; the first one runs normally and fills the runtime's caches) and replayed as one HIP graph afterwards; copyToPixel stays
        outside because its scale argument changes every pass (code.js:1412)."""
        s, k = self.s, self.k
        if use_graph and on_primary is None and self.passes > 1:
            if getattr(self, "_graph", None) is None or self._graph_bounces != bounces:
                if getattr(self, "_graph", None) is not None:
                    self.ctx.graph_release(self._graph)
                self.ctx.capture_begin()
                self._recording = True
                try:
                    self._enqueue_segments(bounces)
                finally:
                    self._recording = False
                    self._graph = self.ctx.capture_end()
                self._graph_bounces = bounces
            self.ctx.graph_launch(self._graph)
        else:
            self._enqueue_segments(bounces, on_primary)
        div = np.float32(1.0 / (s.rpp * self.passes))  # executeCopyToPixel (code.js:1410-1415)
        k["copyToPixel"].set_arg(2, _f32(div)).enqueue(self.gws["copyToPixel"], [64])
        self.ctx.finish()
        self.passes += 1

    def read(self, name):
        from_dt = {"seeds": np.int32, "acu": np.float32, "pixel": np.uint8, "rays": np.uint8, "pois": np.uint8, "shadow": np.uint8}
        return self.b[name].read(from_dt[name])

    def release(self):
        if getattr(self, "_graph", None) is not None:
            self.ctx.graph_release(self._graph)
            self._graph = None
        for k in self.k.values():
            k.release()
        for b in self.b.values():
            b.release()
        self.dev.release()
        self.k, self.b = {}, {}


class FusedRenderer:
    """One mirt_render_pass per progressive pass over rows [row0, row0+nrows)."""

    def __init__(self, ctx, scene, seeds=None, seed_base=0, row0=0, nrows=None, want_radiance=True, keep_acu=True):
        """nrows None: the whole image from row0 = 0.  nrows == 0 is an EMPTY tile (more ranks than rows): it owns minimal buffers and
        its passes do nothing -- it is not the whole frame.
        keep_acu False: no per-ray accumulator at all (16 B per ray never allocated); only a frame's first pass can then run, with
        rays_per_pixel dividing 256, or 256 times a power of two up to 32 (1024, 4096): the pass resolves its pixels itself
        (mirt_render_first_pass with acu == NULL)."""
        self.ctx, self.s = ctx, scene
        self.dev = mirt.DeviceScene(ctx, scene)
        self.row0 = row0
        self.nrows = scene.height if nrows is None else nrows
        self.npix = self.nrows * scene.width
        self.nrays = self.npix * scene.rpp
        self.first_ray = row0 * scene.width * scene.rpp
        self.seeds = ctx.buffer(self.nrays * 4 or 16)
        if self.nrays:
            if seeds is not None:
                self.seeds.write(np.asarray(seeds, np.int32)[self.first_ray:self.first_ray + self.nrays])
            else:
                ctx.seed_fill(self.seeds, self.first_ray, self.nrays, seed_base)
        self.acu = None
        if keep_acu:
            self.acu = ctx.buffer(self.nrays * 16 or 16)
            ctx.zero(self.acu)
        self.pixel = ctx.buffer(self.npix * 4 or 16)
        self.radiance = ctx.buffer(self.npix * 16 or 16) if want_radiance else None
        self.passes = 1
        self._desc = None

    def execute_render(self, bounces=5, fresh=False):
        """fresh: first pass of a frame, accumulator initialised by the pass itself (no ctx.zero needed)."""
        if not self.nrays:
            self.passes += 1
            return
        d = self.dev.pass_desc(self.seeds, self.acu, self.pixel, self.radiance, pass_index=self.passes, bounces=bounces,
                               row0=self.row0, nrows=self.nrows)
        self.ctx.render_pass(d, fresh=fresh)
        self.passes += 1

    def release(self):
        for b in (self.seeds, self.acu, self.pixel, self.radiance):
            if b:
                b.release()
        self.dev.release()


class FramePacked:
    """Packed inputs of an Assign01 / 04 / 07 frame job (what `node host/cli.js pack-frame` emits)."""

    def __init__(self, d):
        self.assign, self.width, self.height = int(d["assign"]), int(d["width"]), int(d["height"])
        self.cam = np.asarray(d["cam"], np.float32)
        self.mol = "atoms" in d       # A07 molecule mode: parsePDB + splitMolData + molTrace (A07 code.js:569-600)
        if self.mol:
            self.bounds = np.asarray(d["bounds"], np.float32)
            self.s_size = int(d["s_size"])
            self.atoms = np.asarray(d["atoms"], np.float32)
            self.mindex, self.mcolor = np.asarray(d["mindex"], np.uint32), np.asarray(d["mcolor"], np.float32)
        elif self.assign != 1:
            self.bounds = np.asarray(d["bounds"], np.float32)
            self.t_size = int(d["t_size"])
            self.pos, self.normal = np.asarray(d["pos"], np.float32), np.asarray(d["normal"], np.float32)
            self.mindex, self.mcolor = np.asarray(d["mindex"], np.uint32), np.asarray(d["mcolor"], np.float32)
        if self.assign == 7:
            self.n_slabs, self.slab_size = int(d["n_slabs"]), np.asarray(d["slab_size"], np.uint32)


def render_frame(ctx, p, timing=None):
    """compute() of A01 (code.js:166-269) / computeTri() of A04 (code.js:553-577) and A07 (code.js:603-628) over the C ABI:
    same kernels, argument indices and NDRange.  Returns (pixels [H*W,4] uint8, rays bytes or None).
    timing: a dict that receives "trace_ms", the HIP-event time of the frame's trace kernel alone (raytrace / meshTrace / molTrace)."""
    def timed(k, gws, l):
        if timing is None:
            return k.enqueue(gws, l)
        ctx.finish()
        ctx.timer_start()
        k.enqueue(gws, l)
        timing["trace_ms"] = ctx.timer_stop_ms()

    pre = {1: "A01:", 4: "A04:", 7: "A07:"}[p.assign]
    w, h = p.width, p.height
    l = get_local_ws(2, 64)
    gws = [-(-w // l[0]) * l[0], -(-h // l[1]) * l[1]]
    pixels = ctx.buffer(w * h * 4, mirt.MEM_WRITE_ONLY)
    keep = [pixels]
    try:
        if p.assign == 1:
            k = ctx.kernel(pre + "raytrace").set_args(pixels, p.cam)
            timed(k, gws, l)
            k.release()
            return pixels.read(np.uint8).reshape(-1, 4), None
        k = ctx.kernel(pre + "sizeofRay")
        tmp = ctx.buffer(4)
        k.set_arg(0, tmp).enqueue([1], [1])
        ray_size = int(tmp.read(np.uint32, 1)[0])
        tmp.release(); k.release()
        rays = ctx.buffer(w * h * ray_size)
        up = lambda a: keep.append(ctx.buffer_from(a)) or keep[-1]
        keep.append(rays)
        it = ctx.kernel(pre + "initTrace").set_args(pixels, p.cam, rays)
        if p.assign == 7:
            it.set_arg(3, p.bounds)
        if p.mol:   # prepareMolTrace / executeMolTrace (A07 code.js:434-470, 549-552): ten arguments, s_mindex / m_color bound but unread
            mt = ctx.kernel(pre + "molTrace").set_args(pixels, p.cam, rays, _u32(p.s_size), up(p.atoms), up(p.mindex), up(p.mcolor), p.bounds,
                                                       _u32(p.n_slabs), up(p.slab_size))
            it.enqueue(gws, l)
            timed(mt, gws, l)
            ctx.finish()
            it.release(); mt.release()
            return pixels.read(np.uint8).reshape(-1, 4), rays.read(np.uint8)
        mt = ctx.kernel(pre + "meshTrace").set_args(pixels, p.cam, rays, _u32(p.t_size), up(p.pos), up(p.normal), up(p.mindex), up(p.mcolor))
        if p.assign == 7:
            mt.set_arg(8, p.bounds).set_arg(9, _u32(p.n_slabs)).set_arg(10, up(p.slab_size))
        it.enqueue(gws, l)
        timed(mt, gws, l)
        ctx.finish()
        it.release(); mt.release()
        return pixels.read(np.uint8).reshape(-1, 4), rays.read(np.uint8)
    finally:
        for b in keep:
            b.release()


def frame_resized(d, w, h):
    """The same frame job at another size: Camera.set (A07 code.js:55-71) makes width = height * cols / rows; cols, rows ride in .sE / .sF."""
    d = dict(d, width=w, height=h)
    cam = list(d["cam"])
    cam[12] = float(np.float32(cam[13] * (w / h)))
    cam[14], cam[15] = float(w), float(h)
    d["cam"] = cam
    return d


def frame_regrid(ctx, a07_job, a04_job, n):
    """The Assign07 job of a mesh at another n_slabs, binned ON THE DEVICE (mirt_grid_build + gathers = splitMeshData, A07 code.js:631-767):
    the mesh's triangles in input order come from its Assign04 job (one slot per triangle), the bounds from the Assign07 job."""
    tri = np.asarray(a04_job["pos"], np.float32).reshape(-1, 3, 4)[:, :, :3].reshape(-1, 9).astype(np.float64)
    nor = np.asarray(a04_job["normal"], np.float32).reshape(-1, 3, 4)[:, :, :3].reshape(-1, 9).astype(np.float64)
    b = np.asarray(a07_job["bounds"], np.float64)
    off, order, total = ctx.grid_build(1, tri, [b[0], b[1], b[2], b[4], b[5], b[6]], n)
    pos, nrm = ctx.grid_gather_triangles(order, total, tri, nor, pad_w=0.0)
    mi = ctx.grid_gather_u32(order, total, np.asarray(a04_job["mindex"], np.uint32))
    out = dict(a07_job, n_slabs=n, slab_size=off.read(np.uint32).tolist(), pos=pos.read(np.float32, total * 12).tolist(),
               normal=nrm.read(np.float32, total * 12).tolist(), mindex=mi.read(np.uint32, total).tolist())
    for buf in (off, order, pos, nrm, mi):
        buf.release()
    return out
