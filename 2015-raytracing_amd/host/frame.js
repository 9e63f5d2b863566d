// host/frame.js -- the single-frame jobs of the reference's earlier assignments (BASELINE configs 1-3):
//   Assign01 compute()      one hard-coded sphere, kernel `raytrace`                (A01 code.js:166-269)
//   Assign04 computeTri()   brute-force mesh, kernels `initTrace`, `meshTrace`      (A04 code.js:553-577, 396-497)
//   Assign07 computeTri()   3-D uniform grid,  kernels `initTrace`, `meshTrace`     (A07 code.js:603-628, 398-557, 980-1122)
//   Assign07 compute()      molecule (.pdb) as spheres in the same grid, kernels `initTrace`, `molTrace`   (A07 code.js:569-600, 434-470, 889-978)
// packFrame() produces the exact typed arrays those hosts upload; FrameRenderer enqueues the kernels in the same
// order through the WebCL-shaped API (./webcl.js), selecting the kernel set by program dialect.
"use strict";
const { webcl } = require("./webcl.js");
const scene = require("./scene.js");
const { parsePDB } = require("./pdb.js");

// what createProgram() gets in place of code.cl: enough text for the runtime to recognise the kernel set
const MANIFEST = {
  1: "__kernel void raytrace(__global uchar4* pixels, float16 fcam);",
  4: "__kernel void sizeofRay(__global uint* s);\n__kernel void initTrace();\n__kernel void meshTrace(uint t_size);",
  7: "__kernel void sizeofRay(__global uint* s);\n__kernel void initTrace();\n__kernel void meshTrace(uint t_size, uint z_stride);\n__kernel void molTrace(uint s_size);",
};

function packFrame(assign, model, width, height, nSlabs) {
  const cam = new scene.Camera();
  if (assign === 1) {  // A01 code.js:180-185; its toFloat32Array packs rows, cols (code.js:50) and W = -z (code.js:57)
    cam.defaultInit(); cam.W = { x: 0, y: 0, z: -1 }; cam.width = 2.66; cam.height = 2.0;
    const f = cam.toFloat32Array(); f[14] = height; f[15] = width;
    return { assign: 1, width: width, height: height, cam: f };
  }
  if (assign === 7 && model && typeof model.pdb === "string") return packMol(parsePDB(model.pdb), cam, width, height, nSlabs);   // { pdb: <file text> }
  const md = scene.parseMeshJSON(model);
  cam.defaultInit(); cam.set(md.bounds, width, height);
  const p = { assign: assign, width: width, height: height, cam: cam.toFloat32Array(), bounds: scene.bounds2AABB(md.bounds),
              t_size: md.nTriangles, mcolor: new Float32Array(md.materials) };
  const T = md.nTriangles;
  if (assign === 4) {  // toPosArray / toNormalArray (A04 code.js:819-870): input order, w = 1 / w = 0 padding
    p.pos = new Float32Array(T * 12); p.normal = new Float32Array(T * 12);
    for (let i = 0; i < T; i++) for (let v = 0; v < 3; v++) {
      for (let c = 0; c < 3; c++) { p.pos[12 * i + 4 * v + c] = md.positions[9 * i + 3 * v + c]; p.normal[12 * i + 4 * v + c] = md.normals[9 * i + 3 * v + c]; }
      p.pos[12 * i + 4 * v + 3] = 1;
    }
    p.mindex = new Uint32Array(md.materialIndices);
  } else {  // splitMeshData with the page's n_slabs (A07 code.js:980-1122): same binning as A10's, plus material indices
    const box = (i) => {
      const q = md.positions, o = 9 * i, mn = (a, b, c) => Math.min(Math.min(a, b), c), mx = (a, b, c) => Math.max(Math.max(a, b), c);
      return [[mn(q[o], q[o + 3], q[o + 6]), mn(q[o + 1], q[o + 4], q[o + 7]), mn(q[o + 2], q[o + 5], q[o + 8])],
              [mx(q[o], q[o + 3], q[o + 6]), mx(q[o + 1], q[o + 4], q[o + 7]), mx(q[o + 2], q[o + 5], q[o + 8])]];
    };
    const g = scene.buildGrid(T, nSlabs, md.bounds, box);
    const n = g.order.length;
    p.pos = new Float32Array(n * 12); p.normal = new Float32Array(n * 12); p.mindex = new Uint32Array(n);
    for (let k = 0; k < n; k++) {
      const i = g.order[k];
      for (let v = 0; v < 3; v++) for (let c = 0; c < 3; c++) {
        p.pos[12 * k + 4 * v + c] = md.positions[9 * i + 3 * v + c]; p.normal[12 * k + 4 * v + c] = md.normals[9 * i + 3 * v + c];
      }
      p.mindex[k] = md.materialIndices[i];
    }
    p.slab_size = g.offsets; p.n_slabs = nSlabs;
  }
  return p;
}

// splitMolData (A07 code.js:889-978): the sphere binning of A10's splitSphereData over (type, x, y, z) records; slots carry
// (c, r*r) and the atom's type index.  The loop runs to `size` (largest serial), past the packed atoms when serials have gaps:
// those reads are undefined -> NaN boxes -> no cell, exactly as in the page.
function packMol(mol, cam, width, height, nSlabs) {
  cam.defaultInit(); cam.set(mol.bounds, width, height);
  const A = mol.atomData, rad = (i) => mol.radiusData[A[4 * i]];
  const g = scene.buildGrid(mol.size, nSlabs, mol.bounds, (i) => {
    const r = rad(i), x = A[4 * i + 1], y = A[4 * i + 2], z = A[4 * i + 3];
    return [[x - r, y - r, z - r], [x + r, y + r, z + r]];
  });
  const n = g.order.length, atoms = new Float32Array(4 * n), mindex = new Uint32Array(n);
  for (let k = 0; k < n; k++) {
    const i = g.order[k], r = rad(i);
    atoms[4 * k] = A[4 * i + 1]; atoms[4 * k + 1] = A[4 * i + 2]; atoms[4 * k + 2] = A[4 * i + 3]; atoms[4 * k + 3] = r * r;
    mindex[k] = A[4 * i];
  }
  return { assign: 7, width: width, height: height, cam: cam.toFloat32Array(), bounds: scene.bounds2AABB(mol.bounds), n_slabs: nSlabs, s_size: mol.size,
           atoms: atoms, mindex: mindex, slab_size: g.offsets, mcolor: new Float32Array(mol.colorData),
           pdb: { size: mol.size, atomData: mol.atomData, colorData: mol.colorData, radiusData: mol.radiusData, min: mol.bounds.min, max: mol.bounds.max } };
}

const ceilTo = (n, m) => Math.ceil(n / m) * m;

function renderFrame(p, opt) {
  opt = opt || {};
  const devs = webcl.getPlatforms()[0].getDevices(webcl.DEVICE_TYPE_ALL);
  if (!devs.length) throw new Error("no MI355X visible: this host has no CPU path");
  const device = devs[opt.device || 0];
  const ctx = webcl.createContext(device), q = ctx.createCommandQueue();
  const prog = ctx.createProgram(MANIFEST[p.assign]);
  prog.build();
  const w = p.width, h = p.height, res = [];
  const buf = (flags, a) => { const b = ctx.createBuffer(flags, Math.max(a.byteLength || a, 16)); res.push(b); if (a.byteLength) q.enqueueWriteBuffer(b, false, 0, a.byteLength, a, []); return b; };
  const pixels = buf(webcl.MEM_WRITE_ONLY, w * h * 4);
  const gws = [ceilTo(w, 8), ceilTo(h, 8)], lws = [8, 8];   // getLocalWS(2, 64)
  if (p.assign === 1) {
    const k = prog.createKernel("raytrace"); res.push(k);
    k.setArg(0, pixels); k.setArg(1, p.cam);
    q.enqueueNDRangeKernel(k, 2, null, gws, lws);
  } else {
    const sk = prog.createKernel("sizeofRay"), sb = ctx.createBuffer(webcl.MEM_WRITE_ONLY, 4), so = new Uint32Array(1);   // getRaySize
    sk.setArg(0, sb); q.enqueueNDRangeKernel(sk, 1, null, [1], [1]); q.enqueueReadBuffer(sb, false, 0, 4, so, []); q.finish(); sb.release(); sk.release();
    const rays = buf(webcl.MEM_READ_WRITE, w * h * so[0]);
    const it = prog.createKernel("initTrace"), mt = prog.createKernel(p.atoms ? "molTrace" : "meshTrace"); res.push(it, mt);
    it.setArg(0, pixels); it.setArg(1, p.cam); it.setArg(2, rays);
    if (p.assign === 7) it.setArg(3, p.bounds);
    const RO = webcl.MEM_READ_ONLY;
    if (p.atoms) {   // prepareMolTrace (A07 code.js:434-470): ten arguments
      [pixels, p.cam, rays, new Uint32Array([p.s_size]), buf(RO, p.atoms), buf(RO, p.mindex), buf(RO, p.mcolor), p.bounds, new Uint32Array([p.n_slabs]),
       buf(RO, p.slab_size)].forEach((v, i) => mt.setArg(i, v));
    } else [pixels, p.cam, rays, new Uint32Array([p.t_size]), buf(RO, p.pos), buf(RO, p.normal), buf(RO, p.mindex), buf(RO, p.mcolor)].forEach((v, i) => mt.setArg(i, v));
    if (p.assign === 7 && !p.atoms) { mt.setArg(8, p.bounds); mt.setArg(9, new Uint32Array([p.n_slabs])); mt.setArg(10, buf(RO, p.slab_size)); }
    q.enqueueNDRangeKernel(it, 2, null, gws, lws);
    q.enqueueNDRangeKernel(mt, 2, null, gws, lws);
  }
  const out = new Uint8ClampedArray(w * h * 4);
  q.enqueueReadBuffer(pixels, false, 0, out.length, out, []);
  q.finish();
  res.forEach((r) => r.release());
  prog.release(); q.release(); ctx.release();
  return out;
}

module.exports = { packFrame, packMol, renderFrame, MANIFEST };
