// host/renderer.js -- the render drivers of the JavaScript host.
//
// GranularRenderer is the reference's kernel plumbing, call for call: preRender() and
// executeRender() of A10 code.js:1784-1854 with the prepare*/execute* helpers (:1078-1528),
// written against the WebCL-shaped API of ./webcl.js -- same kernels, same argument indices,
// same enqueue order, same NDRange padding.  FusedRenderer asks the runtime for the same pass
// in one launch (queue.renderPass).  Both produce bit-identical frames.
"use strict";
let { webcl } = require("./webcl.js");
const scene = require("./scene.js");
// tests point the drivers at another implementation of the same object model (./webcl_record.js) to compare call streams
function setWebCL(impl) { webcl = impl; }

const KERNELS = ["sizeofRay", "sizeofPoi", "initAcu", "initTrace", "sphereTrace", "triangleTrace", "meshTrace", "lightRender",
                 "initShadowTrace", "sphereShadowTrace", "triangleShadowTrace", "sceneRender", "bouncePaths", "copyToPixel"];
// what createProgram() is given in place of code.cl: the list of kernels this host will ask for
const MANIFEST = KERNELS.map((k) => "__kernel void " + k + "();").join("\n");

const u32 = (v) => new Uint32Array([v]);
const f32 = (v) => new Float32Array([v]);
const ceilTo = (n, m) => Math.ceil(n / m) * m;

function getLocalWS(dim, kernel, device) {  // code.js:645-672
  const m = kernel.getWorkGroupInfo(device, webcl.KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE);
  if (dim === 1) return [m];
  let x = Math.floor(Math.sqrt(m));
  if (x & (x - 1)) { --x; for (let i = 1; i < 32; i <<= 1) x |= x >> i; ++x; }
  return [x, Math.floor(m / x)];
}

function pickDevice(index) {
  const devs = webcl.getPlatforms()[0].getDevices(webcl.DEVICE_TYPE_ALL);
  if (!devs.length) throw new Error("no MI355X visible: this host has no CPU path");
  return devs[index || 0];
}

// uploads the packed scene (scene.packScene) the way prepare*() do (code.js:1175-1185, 1221-1234, 1268-1278, 1375-1379)
function uploadScene(ctx, q, p) {
  const up = (a) => { const b = ctx.createBuffer(webcl.MEM_READ_ONLY, Math.max(a.byteLength, 16)); if (a.byteLength) q.enqueueWriteBuffer(b, false, 0, a.byteLength, a, []); return b; };
  const d = { bufs: [] };
  const keep = (b) => { d.bufs.push(b); return b; };
  if (p.n_spheres > 0) d.sph = { prims: keep(up(p.spheres)), matid: keep(up(p.s_matid)), cellOffsets: keep(up(p.s_box)), bounds: p.sphere_bounds, nSlabs: p.n_slabs };
  if (p.n_triangles > 0) d.tri = { prims: keep(up(p.t_pos)), normals: keep(up(p.t_normal)), matid: keep(up(p.t_matid)), cellOffsets: keep(up(p.t_box)), bounds: p.triangle_bounds, nSlabs: p.n_slabs };
  d.meshes = p.meshes.map((m) => ({ prims: keep(up(m.pos)), normals: keep(up(m.normal)), cellOffsets: keep(up(m.box)), bounds: m.bounds, nSlabs: m.nslabs, meshMatId: m.matid }));
  d.material = keep(up(p.materials));
  return d;
}

// The same device-side scene as uploadScene(), but binned ON THE DEVICE from the raw primitives (queue.gridBuild /
// gridGather*): `sc` is a scene loaded with {deferGrids: true}, `p` its packed header (camera, lights, bounds, materials).
function buildSceneOnDevice(ctx, q, sc, p) {
  const d = { bufs: [] };
  const keep = (b) => { d.bufs.push(b); return b; };
  const b6 = (b) => [b.min[0], b.min[1], b.min[2], b.max[0], b.max[1], b.max[2]];
  if (sc.spheres.length) {
    const f = new Float64Array(sc.spheres.length * 4);
    sc.spheres.forEach((s, i) => f.set([s.c.x, s.c.y, s.c.z, s.r], 4 * i));
    const g = q.gridBuild(0, f, b6(sc.sphereBounds), p.n_slabs);
    d.sph = { prims: keep(q.gridGatherSpheres(g.order, g.total, f)), matid: keep(q.gridGatherU32(g.order, g.total, new Uint32Array(sc.spheres.map((s) => s.matId)))),
              cellOffsets: keep(g.offsets), bounds: p.sphere_bounds, nSlabs: p.n_slabs };
    g.order.release();
  }
  const tri9 = (T, key) => { const f = new Float64Array(T.length * 9); T.forEach((t, i) => f.set([t[key[0]].x, t[key[0]].y, t[key[0]].z, t[key[1]].x, t[key[1]].y, t[key[1]].z, t[key[2]].x, t[key[2]].y, t[key[2]].z], 9 * i)); return f; };
  if (sc.triangles.length) {
    const pos = tri9(sc.triangles, ["p0", "p1", "p2"]), nor = tri9(sc.triangles, ["n0", "n1", "n2"]);
    const g = q.gridBuild(1, pos, b6(sc.triangleBounds), p.n_slabs);
    const t = q.gridGatherTriangles(g.order, g.total, pos, nor, [], 0);
    d.tri = { prims: keep(t.pos), normals: keep(t.nor), matid: keep(q.gridGatherU32(g.order, g.total, new Uint32Array(sc.triangles.map((x) => x.matId)))),
              cellOffsets: keep(g.offsets), bounds: p.triangle_bounds, nSlabs: p.n_slabs };
    g.order.release();
  }
  d.meshes = sc.meshes.map((m, i) => {
    // the soups are on the device already (queue.meshIngest = parseMeshJSON there) -- on the context that loaded the scene.  A row tile
    // on ANOTHER context (renderTiled: one context per device) ingests the mesh file again on its own device: buffers do not cross contexts.
    const jm = m.jmesh;
    let onDev = !!jm.positionsBuf, own = false, pos, nor;
    if (onDev && jm.positionsBuf.ctx !== ctx) {
      const r = q.meshIngest(jm.model, scene.normalFromMat4);
      pos = r.positionsBuf; nor = r.normalsBuf; own = true;
    } else if (onDev) { pos = jm.positionsBuf; nor = jm.normalsBuf; }
    else { pos = new Float64Array(jm.positions); nor = new Float64Array(jm.normals); }
    const g = q.gridBuild(1, pos, b6(m.gridBounds), m.nslabs, jm.nTriangles);      // grid on the untransformed mesh (code.js:106-112)
    const t = q.gridGatherTriangles(g.order, g.total, pos, nor, m.steps, 0);   // normalize / scale / translate in fp64, then fp32
    g.order.release();
    if (own) { pos.release(); nor.release(); }
    else if (onDev && !sc.keepSoups) { pos.release(); nor.release(); jm.positionsBuf = jm.normalsBuf = null; }
    return { prims: keep(t.pos), normals: keep(t.nor), cellOffsets: keep(g.offsets), bounds: p.meshes[i].bounds, nSlabs: m.nslabs, meshMatId: m.matId };
  });
  const mat = ctx.createBuffer(webcl.MEM_READ_ONLY, Math.max(p.materials.byteLength, 16));
  q.enqueueWriteBuffer(mat, false, 0, p.materials.byteLength, p.materials, []);
  d.material = keep(mat);
  return d;
}

class GranularRenderer {
  constructor(packed, opt) {
    opt = opt || {};
    this.p = packed;
    this.ownCtx = !opt.ctx;
    this.device = opt.ctx ? opt.ctx.device : pickDevice(opt.device);
    this.useGraph = !!opt.graph;
    this.ctx = opt.ctx || webcl.createContext(this.device);           // createCLBasicResources (code.js:576-608)
    // ours: with `fusion` the runtime runs each pass of this stream as one fused launch (webcl.createContext's default); without it this
    // renderer is the launch-by-launch path on purpose -- whoever made the context (renderFile's --device-grid hands one in)
    if (this.ctx.setFusion) this.ctx.setFusion(opt.fusion ? 2 : 0);
    this.q = this.ctx.createCommandQueue();
    this.program = this.ctx.createProgram(MANIFEST);
    this.program.build();
    this.k = {}; this.b = {}; this.gws = {}; this.lws = {};
    this.passes = 1;
    this.sceneObject = opt.sceneObject || null;
    this._preRender(opt.seeds, opt.seedBase);
  }
  _structSize(name) {  // getStructSize (code.js:1064-1076)
    const k = this.program.createKernel("sizeof" + name), b = this.ctx.createBuffer(webcl.MEM_WRITE_ONLY, 4), out = new Uint32Array(1);
    k.setArg(0, b);
    this.q.enqueueNDRangeKernel(k, 1, null, [1], [1]);
    this.q.enqueueReadBuffer(b, false, 0, 4, out, []);
    this.q.finish();
    b.release(); k.release();
    return out[0];
  }
  _preRender(seeds, seedBase) {
    const p = this.p, ctx = this.ctx, q = this.q, prog = this.program, k = this.k, b = this.b;
    const n = this.totalRays = p.rays_per_pixel * p.width * p.height;
    // prepareInitAcu
    b.acu = ctx.createBuffer(webcl.MEM_READ_WRITE, n * 16);
    const ia = prog.createKernel("initAcu");
    ia.setArg(0, b.acu); ia.setArg(1, u32(n));
    let l = getLocalWS(1, ia, this.device);
    q.enqueueNDRangeKernel(ia, 1, null, [ceilTo(n, l[0])], l);
    q.finish(); ia.release();
    // prepareInitSeeds: Math.random() in the reference; a caller-supplied array or the device-side closed form here
    b.seeds = ctx.createBuffer(webcl.MEM_READ_WRITE, n * 4);
    if (seeds) q.enqueueWriteBuffer(b.seeds, false, 0, n * 4, seeds, []); else q.seedFill(b.seeds, 0, n, seedBase || 0);
    // prepareInitTrace
    const raySize = this._structSize("Ray"), poiSize = this._structSize("Poi");
    b.rays = ctx.createBuffer(webcl.MEM_READ_WRITE, n * raySize);
    b.pois = ctx.createBuffer(webcl.MEM_READ_WRITE, n * poiSize);
    k.initTrace = prog.createKernel("initTrace");
    [b.seeds, b.rays, b.pois, p.bounds].forEach((v, i) => k.initTrace.setArg(i, v));
    k.initTrace.setArg(5, f32(p.focal_length)); k.initTrace.setArg(6, f32(p.lens_rad)); k.initTrace.setArg(7, u32(p.rays_per_pixel));
    l = this.lws.initTrace = getLocalWS(2, k.initTrace, this.device);
    this.gws.initTrace = [ceilTo(p.width, l[0]), ceilTo(p.height, l[1])];
    const d = this.dev = this.sceneObject ? buildSceneOnDevice(ctx, q, this.sceneObject, p) : uploadScene(ctx, q, p);
    if (d.sph) {  // prepareSphereTrace
      k.sphereTrace = prog.createKernel("sphereTrace");
      [u32(n), b.pois, b.rays, d.sph.prims, d.sph.matid, d.sph.cellOffsets, d.sph.bounds, u32(d.sph.nSlabs)].forEach((v, i) => k.sphereTrace.setArg(i, v));
    }
    if (d.tri) {  // prepareTriangleTrace
      k.triangleTrace = prog.createKernel("triangleTrace");
      [u32(n), b.pois, b.rays, d.tri.prims, d.tri.normals, d.tri.matid, d.tri.cellOffsets, d.tri.bounds, u32(d.tri.nSlabs)].forEach((v, i) => k.triangleTrace.setArg(i, v));
    }
    if (d.meshes.length) {  // prepareMeshTrace
      k.meshTrace = prog.createKernel("meshTrace");
      [u32(n), b.pois, b.rays].forEach((v, i) => k.meshTrace.setArg(i, v));
    }
    // prepareInitShadowTrace: asks for the Ray size again (code.js:1417-1419)
    b.shadow = ctx.createBuffer(webcl.MEM_READ_WRITE, n * this._structSize("Ray"));
    k.initShadowTrace = prog.createKernel("initShadowTrace");
    k.initShadowTrace.setArg(0, b.shadow); k.initShadowTrace.setArg(1, b.pois); k.initShadowTrace.setArg(2, u32(n)); k.initShadowTrace.setArg(4, b.seeds);
    if (d.sph) {
      k.sphereShadowTrace = prog.createKernel("sphereShadowTrace");
      [u32(n), b.shadow, d.sph.prims, d.sph.cellOffsets, d.sph.bounds, u32(d.sph.nSlabs)].forEach((v, i) => k.sphereShadowTrace.setArg(i, v));
    }
    if (d.tri || d.meshes.length) {
      k.triangleShadowTrace = prog.createKernel("triangleShadowTrace");
      k.triangleShadowTrace.setArg(0, u32(n)); k.triangleShadowTrace.setArg(1, b.shadow);
    }
    // prepareSceneRender / prepareCopyToPixel / prepareBouncePaths / prepareLightRender
    k.sceneRender = prog.createKernel("sceneRender");
    [b.acu, b.pois, b.shadow, d.material].forEach((v, i) => k.sceneRender.setArg(i, v));
    k.sceneRender.setArg(5, u32(n));
    const npix = p.width * p.height;
    b.pixel = ctx.createBuffer(webcl.MEM_WRITE_ONLY, npix * 4);
    k.copyToPixel = prog.createKernel("copyToPixel");
    k.copyToPixel.setArg(0, b.pixel); k.copyToPixel.setArg(1, b.acu); k.copyToPixel.setArg(3, u32(npix)); k.copyToPixel.setArg(4, u32(p.rays_per_pixel));
    this.gws.copyToPixel = [ceilTo(npix, 64)];
    k.bouncePaths = prog.createKernel("bouncePaths");
    [b.pois, b.rays, b.seeds, u32(n)].forEach((v, i) => k.bouncePaths.setArg(i, v));
    k.lightRender = prog.createKernel("lightRender");
    k.lightRender.setArg(0, b.pois); k.lightRender.setArg(1, b.rays); k.lightRender.setArg(2, b.acu); k.lightRender.setArg(4, u32(n));
    this.g1 = [ceilTo(n, 64)];
  }
  _run(kernel) { this.q.enqueueNDRangeKernel(kernel, 1, null, this.g1, [64]); }
  _closest() {
    const k = this.k, d = this.dev;
    if (d.sph) this._run(k.sphereTrace);
    if (d.tri) this._run(k.triangleTrace);
    for (const m of d.meshes) {  // executeMeshTrace (code.js:1293-1303)
      k.meshTrace.setArg(3, m.prims); k.meshTrace.setArg(4, m.normals); k.meshTrace.setArg(5, m.cellOffsets);
      k.meshTrace.setArg(6, u32(m.meshMatId)); k.meshTrace.setArg(7, m.bounds); k.meshTrace.setArg(8, u32(m.nSlabs));
      this._run(k.meshTrace);
    }
  }
  _direct() {
    const k = this.k, d = this.dev;
    for (const light of this.p.lights) {
      k.initShadowTrace.setArg(3, light.shadow); this._run(k.initShadowTrace);
      if (d.sph) this._run(k.sphereShadowTrace);
      const sets = (d.tri ? [d.tri] : []).concat(d.meshes);  // executeTriangleShadowTrace / executeMeshShadowTrace (code.js:1514-1528)
      for (const s of sets) {
        k.triangleShadowTrace.setArg(2, s.prims); k.triangleShadowTrace.setArg(3, s.cellOffsets); k.triangleShadowTrace.setArg(4, s.bounds);
        k.triangleShadowTrace.setArg(5, u32(s.nSlabs)); this._run(k.triangleShadowTrace);
      }
      k.sceneRender.setArg(4, light.scene); this._run(k.sceneRender);  // executeSceneRender (code.js:1402-1408)
      if (!this._recording) this.q.finish();   // the reference finishes the queue here (code.js:1406); a recording cannot wait
    }
  }
  _segments(bounces) {
    const k = this.k, p = this.p;
    k.initTrace.setArg(4, p.cam);
    this.q.enqueueNDRangeKernel(k.initTrace, 2, null, this.gws.initTrace, this.lws.initTrace);
    this._closest();
    for (const light of p.lights) { k.lightRender.setArg(3, light.light); this._run(k.lightRender); }
    this._direct();
    for (let j = 0; j < bounces; j++) { this._run(k.bouncePaths); this._closest(); this._direct(); }
  }
  // code.js:1806-1854.  opt.graph (constructor): from the second pass on the pass body -- 40-odd launches, each a few microseconds on
  // the page's 320x240 canvas -- is one HIP graph replay; copyToPixel stays outside, its scale changes every pass (code.js:1412)
  executeRender(bounces) {
    const k = this.k, p = this.p;
    bounces = bounces === undefined ? 5 : bounces;
    if (this.useGraph && this.passes > 1) {
      if (!this._graph || this._graphBounces !== bounces) {
        if (this._graph) this._graph.release();
        this.q.captureBegin();
        this._recording = true;
        try { this._segments(bounces); } finally { this._recording = false; this._graph = this.q.captureEnd(); }
        this._graphBounces = bounces;
      }
      this.q.launchGraph(this._graph);
    } else this._segments(bounces);
    k.copyToPixel.setArg(2, f32(1.0 / (p.rays_per_pixel * this.passes)));  // executeCopyToPixel (code.js:1410-1415)
    this.q.enqueueNDRangeKernel(k.copyToPixel, 1, null, this.gws.copyToPixel, [64]);
    this.passes++;
  }
  readPixels() {  // sendImagetoHTML (code.js:1530-1537)
    const out = new Uint8ClampedArray(this.p.width * this.p.height * 4);
    this.q.enqueueReadBuffer(this.b.pixel, false, 0, out.length, out, []);
    this.q.finish();
    return out;
  }
  readAcu() { const a = new Float32Array(this.totalRays * 4); this.q.enqueueReadBuffer(this.b.acu, false, 0, a.byteLength, a, []); this.q.finish(); return a; }
  release() {  // releaseCLResources (code.js:1539-1552)
    if (this._graph) { this._graph.release(); this._graph = null; }
    Object.values(this.k).forEach((k) => k.release());
    Object.values(this.b).forEach((b) => b.release());
    this.dev.bufs.forEach((b) => b.release());
    this.program.release(); this.q.release();
    if (this.ownCtx) this.ctx.release();
  }
}

class FusedRenderer {
  constructor(packed, opt) {
    opt = opt || {};
    const p = this.p = packed;
    this.ownCtx = !opt.ctx;
    this.device = opt.ctx ? opt.ctx.device : pickDevice(opt.device);
    this.ctx = opt.ctx || webcl.createContext(this.device);   // opt.ctx: a context of a device group (renderTiled)
    this.q = this.ctx.createCommandQueue();
    this.row0 = opt.row0 || 0;
    this.nrows = opt.nrows === undefined ? p.height : opt.nrows;
    this.npix = this.nrows * p.width;
    this.nrays = this.npix * p.rays_per_pixel;
    const first = this.row0 * p.width * p.rays_per_pixel;
    this.dev = opt.sceneObject ? buildSceneOnDevice(this.ctx, this.q, opt.sceneObject, p) : uploadScene(this.ctx, this.q, p);
    this.seeds = this.ctx.createBuffer(webcl.MEM_READ_WRITE, this.nrays * 4);
    if (opt.seeds) this.q.enqueueWriteBuffer(this.seeds, false, 0, this.nrays * 4, opt.seeds.subarray(first, first + this.nrays), []);
    else this.q.seedFill(this.seeds, first, this.nrays, opt.seedBase || 0);
    // not zeroed: the first pass initialises it (firstPass below).  opt.keepAcu === false: a one-pass frame without the 16 bytes per ray -- the pass
    // resolves its pixels itself (mirt_render_first_pass with acu == NULL: rays_per_pixel must divide 256 or be 256 times a power of two up to 32, and there is no second pass)
    this.acu = opt.keepAcu === false ? null : this.ctx.createBuffer(webcl.MEM_READ_WRITE, this.nrays * 16);
    this.pixel = this.ctx.createBuffer(webcl.MEM_WRITE_ONLY, this.npix * 4);
    this.radiance = this.ctx.createBuffer(webcl.MEM_WRITE_ONLY, this.npix * 16);
    this.passes = 1;
  }
  executeRender(bounces) {
    const p = this.p, d = this.dev;
    this.q.renderPass({ width: p.width, height: p.height, raysPerPixel: p.rays_per_pixel, row0: this.row0, nrows: this.nrows,
      bounces: bounces === undefined ? 5 : bounces, passIndex: this.passes, cam: p.cam, sceneBounds: p.bounds,
      focalLength: p.focal_length, lensRad: p.lens_rad, spheres: d.sph, triangles: d.tri, meshes: d.meshes, lights: p.lights,
      material: d.material, seeds: this.seeds, acu: this.acu, pixel: this.pixel, radiance: this.radiance,
      firstPass: this.passes === 1 });   // preRender's initAcu (code.js:1078-1099) folded into the frame's first pass
    this.passes++;
  }
  readPixels() { const o = new Uint8ClampedArray(this.npix * 4); this.q.enqueueReadBuffer(this.pixel, false, 0, o.length, o, []); this.q.finish(); return o; }
  readRadiance() { const o = new Float32Array(this.npix * 4); this.q.enqueueReadBuffer(this.radiance, false, 0, o.byteLength, o, []); this.q.finish(); return o; }
  readAcu() { const a = new Float32Array(this.nrays * 4); this.q.enqueueReadBuffer(this.acu, false, 0, a.byteLength, a, []); this.q.finish(); return a; }
  release() {
    [this.seeds, this.acu, this.pixel, this.radiance].forEach((b) => b && b.release());
    this.dev.bufs.forEach((b) => b.release());
    this.q.release();
    if (this.ownCtx) this.ctx.release();
  }
}

// One frame over N devices of this process: contiguous row tiles (mirt_tile_rows), global ray ids, every device runs the fused pass on
// its tile (launches are asynchronous: the one JS thread queues them all, the devices run concurrently), then the RGBA8 and radiance
// tiles are gathered on device 0 (group.gather: RCCL over xGMI for N > 1) and read back once.  The image is identical for every N.
function renderTiled(packed, nDevices, passes, opt) {
  opt = opt || {};
  const all = webcl.getPlatforms()[0].getDevices(webcl.DEVICE_TYPE_ALL);
  // rehearsal (MIRT_GROUP_ALLOW_REPEATED_DEVICES=1, include/mirt.h): more tiles than devices -- the contexts share the devices at hand
  const rehearse = process.env.MIRT_GROUP_ALLOW_REPEATED_DEVICES === "1";
  if (all.length < nDevices && !rehearse) throw new Error("renderTiled: " + nDevices + " devices asked for, " + all.length + " visible");
  if (packed.rays_per_pixel === 1 && nDevices > 1) throw new Error("one ray per pixel couples the rows through seeds[col] (A10 code.cl:429): render it on one device");
  const devs = [];
  for (let i = 0; i < nDevices; i++) devs.push(all[i % all.length]);
  const group = webcl.createDeviceGroup(devs);
  const tiles = [];
  for (let i = 0; i < nDevices; i++) {
    const t = group.tileRows(packed.height, i);
    tiles.push(t.nrows ? new FusedRenderer(packed, Object.assign({}, opt, { ctx: group.contexts[i], row0: t.row0, nrows: t.nrows })) : null);
  }
  const live = tiles.filter((t) => t);
  const q0 = live[0].q;
  q0.timerStart();
  for (let p = 0; p < passes; p++) live.forEach((t) => t.executeRender(opt.bounces));
  const npix = packed.width * packed.height;
  const root = group.contexts[0];
  const frame = root.createBuffer(webcl.MEM_READ_WRITE, npix * 4), rad = root.createBuffer(webcl.MEM_READ_WRITE, npix * 16);
  const dummy = tiles.map((t, i) => t || { pixel: group.contexts[i].createBuffer(webcl.MEM_READ_WRITE, 16), radiance: group.contexts[i].createBuffer(webcl.MEM_READ_WRITE, 16), npix: 0 });
  group.gather(dummy.map((t) => t.pixel), dummy.map((t) => t.npix * 4), frame, 0, opt.forceRccl);
  group.gather(dummy.map((t) => t.radiance), dummy.map((t) => t.npix * 16), rad, 0, opt.forceRccl);
  group.finish();
  const ms = q0.timerStopMs();
  // what the gather did: the route of every tile and whether each context reads the root's device directly (mirt.h MIRT_ROUTE_*)
  const routes = group.routes(), peers = tiles.map((_, i) => group.peerAccess(0, i));
  const pixel = new Uint8ClampedArray(npix * 4), radiance = new Float32Array(npix * 4);
  q0.enqueueReadBuffer(frame, true, 0, pixel.length, pixel, []);
  q0.enqueueReadBuffer(rad, true, 0, radiance.byteLength, radiance, []);
  q0.finish();
  const res = { pixel: pixel, radiance: radiance, ms: ms, device: live.length + " x " + live[0].device.getInfo(webcl.DEVICE_NAME), tiles: tiles.map((t) => (t ? [t.row0, t.nrows] : [0, 0])), routes: routes, peerAccess: peers };
  live.forEach((t) => t.release());
  group.release();
  return res;
}

// per-pixel sequential fp32 sums of the per-ray accumulators (copyToPixel's order, A10 code.cl:1377-1380)
function radianceSums(acu, rpp) {
  const n = acu.length / 4 / rpp, out = new Float32Array(n * 4);
  for (let px = 0; px < n; px++) for (let i = 0; i < rpp; i++) for (let c = 0; c < 4; c++) out[4 * px + c] = Math.fround(out[4 * px + c] + acu[4 * (px * rpp + i) + c]);
  return out;
}

function renderFile(file, width, height, rpp, passes, opt) {
  opt = opt || {};
  let packed, ownCtx = null;
  if (opt.deviceGrid) {   // the host only reads the files: mesh ingest (node x mesh transforms, de-indexing: parseMeshJSON), binning, mesh
                          // transforms and fp32 narrowing all happen on the device
    ownCtx = webcl.createContext(pickDevice(opt.device));
    const q = ownCtx.createCommandQueue();
    const parseMesh = (model) => {
      const r = q.meshIngest(model, scene.normalFromMat4);
      r.model = model;   // a row tile on another device ingests it again there (buildSceneOnDevice)
      const b = r.bounds6.map((v) => (v === Infinity ? Number.MAX_VALUE : v === -Infinity ? -Number.MAX_VALUE : v));   // an empty mesh keeps Bounds' initial values
      r.bounds = new scene.Bounds(b.slice(0, 3), b.slice(3));
      return r;
    };
    const sc = scene.loadSceneFile(file, width, height, { deferGrids: true, parseMesh: opt.hostIngest ? undefined : parseMesh });
    packed = scene.packScene(sc, width, height, rpp, 1, true);
    opt = Object.assign({}, opt, { sceneObject: sc, ctx: ownCtx });
  } else packed = scene.packScene(scene.loadSceneFile(file, width, height), width, height, rpp);
  if (opt.gpus) {
    // N contexts, one per device: each builds its own copy of the scene (the loader's soups stay on ownCtx until every tile is built)
    if (opt.sceneObject) opt.sceneObject.keepSoups = true;
    try { return renderTiled(packed, opt.gpus, passes, opt); } finally { if (ownCtx) ownCtx.release(); }
  }
  const R = opt.granular ? new GranularRenderer(packed, opt) : new FusedRenderer(packed, opt);
  R.q.timerStart();
  for (let i = 0; i < passes; i++) R.executeRender(opt.bounces);
  const ms = R.q.timerStopMs();
  const res = { pixel: R.readPixels(), radiance: opt.granular ? radianceSums(R.readAcu(), rpp) : R.readRadiance(), ms: ms,
                device: R.device.getInfo(webcl.DEVICE_NAME), fusedPasses: R.ctx.fusedPasses ? R.ctx.fusedPasses() : 0 };
  R.release();
  if (ownCtx) ownCtx.release();
  return res;
}

module.exports = { GranularRenderer, FusedRenderer, renderFile, renderTiled, radianceSums, getLocalWS, KERNELS, setWebCL };
