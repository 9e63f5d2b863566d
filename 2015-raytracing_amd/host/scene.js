// host/scene.js -- scene ingest for the MI355X path tracer: everything the reference host does
// between "scene file on disk" and "typed arrays handed to the device".
//
// Mirrors, function for function, the data-preparation half of the reference's
// Assign10-Path_Tracing/code.js ("A10 code.js") and tri/meshDataVersion1.js so that its
// scenes/*.xml and tri/*.json load UNCHANGED and every uploaded buffer is bit-identical:
//   Camera.lookAt / toFloat32Array        A10 code.js:203-217, 250-258
//   Light area / TBN / three packings     A10 code.js:298-352
//   loadScene                             A10 code.js:723-897
//   Mesh normalize / scale / translate    A10 code.js:114-169
//   splitSphereData / splitTriangleData / splitMeshData   A10 code.js:1554-1772, 899-1041
//   bounds2AABB, splitMaterialData        A10 code.js:610-621, 1774-1782
//   parseMeshJSON                         A10 tri/meshDataVersion1.js:12-78
//   Bounds {merge, center, diagonal}      A10 lib/utilities.js:389-422 (semantics only)
// Own code throughout: a small XML reader instead of the browser DOM, counting-sort grid
// builders instead of nested arrays, the five gl-matrix operations the mesh parser needs
// restated here.  What must NOT change is the arithmetic: all of it runs in JS doubles in the
// reference's order and narrows to fp32 exactly where the reference does (`new Float32Array`,
// and gl-matrix's Float32Array-backed vec3/mat3/mat4 results).
"use strict";
const fs = require("fs");
const path = require("path");

// ---------------------------------------------------------------------------------------------
// Bounds (utilities.js:389-422)
function Bounds(min, max) {
  this.min = [Number.MAX_VALUE, Number.MAX_VALUE, Number.MAX_VALUE];
  this.max = [-Number.MAX_VALUE, -Number.MAX_VALUE, -Number.MAX_VALUE];
  if (min) this.min = [min[0], min[1], min[2]];
  if (max) this.max = [max[0], max[1], max[2]];
}
Bounds.prototype.merge = function (b) {
  for (let i = 0; i < 3; i++) this.min[i] = Math.min(this.min[i], b.min[i]);
  for (let i = 0; i < 3; i++) this.max[i] = Math.max(this.max[i], b.max[i]);
};
Bounds.prototype.center = function () {
  return [(this.min[0] + this.max[0]) / 2, (this.min[1] + this.max[1]) / 2, (this.min[2] + this.max[2]) / 2];
};
Bounds.prototype.diagonal = function () {
  const dx = this.max[0] - this.min[0], dy = this.max[1] - this.min[1], dz = this.max[2] - this.min[2];
  return Math.sqrt(dx * dx + dy * dy + dz * dz);
};

// bounds2AABB (code.js:610-621): (min,1,max,1) as fp32
function bounds2AABB(b) {
  return new Float32Array([b.min[0], b.min[1], b.min[2], 1, b.max[0], b.max[1], b.max[2], 1]);
}

// ---------------------------------------------------------------------------------------------
// tiny vector helpers on {x,y,z} (code.js:13-53)
const v3 = (x, y, z) => ({ x: x, y: y, z: z });
const sub = (a, b) => v3(a.x - b.x, a.y - b.y, a.z - b.z);
const cross = (a, b) => v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
function normalize(a) {
  const len = Math.sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  a.x /= len; a.y /= len; a.z /= len;
  return a;
}

// Camera (code.js:175-277)
function Camera() {
  this.eye = v3(0, 0, 0); this.U = v3(0, 0, 0); this.V = v3(0, 0, 0); this.W = v3(0, 0, 0);
  this.width = 1.0; this.height = 1.0; this.cols = 0; this.rows = 0;
}
Camera.prototype.lookAt = function (eye, lookat, vup, fov, cols, rows) {
  this.cols = cols; this.rows = rows;
  const aspect = cols / rows;
  this.height = 2.0 * Math.tan(0.5 * fov * Math.PI / 180.0);
  this.width = this.height * aspect;
  this.eye = eye;
  this.W = normalize(sub(eye, lookat));
  this.U = normalize(cross(vup, this.W));
  this.V = cross(this.W, this.U);
};
// the bounds-framing camera of A04/A07 (A07 code.js:55-71; same text in A10 code.js:185-201)
Camera.prototype.defaultInit = function () {
  this.eye = v3(0, 0, 0); this.U = v3(1, 0, 0); this.V = v3(0, 1, 0); this.W = v3(0, 0, 1);
};
Camera.prototype.set = function (bounds, cols, rows) {
  this.cols = cols; this.rows = rows;
  const fov = 60, aspect = cols / rows, c = bounds.center(), diag = bounds.diagonal();
  this.eye.x = c[0]; this.eye.y = c[1]; this.eye.z = c[2] + diag;
  this.height = 2.0 * Math.tan(0.5 * fov * Math.PI / 180.0);
  this.width = this.height * aspect;
};
Camera.prototype.toFloat32Array = function () {
  return new Float32Array([this.eye.x, this.eye.y, this.eye.z, this.U.x, this.U.y, this.U.z, this.V.x, this.V.y, this.V.z,
    this.W.x, this.W.y, this.W.z, this.width, this.height, this.cols, this.rows]);
};

// Light (code.js:279-353).  loadScene assigns the normal as read -- it is NOT normalised there.
function Light() {
  this.position = v3(0, 0, 0); this.normal = v3(0, 0, 0); this.T = v3(0, 0, 0); this.B = v3(0, 0, 0);
  this.irradiance = v3(0, 0, 0); this.radius = 0.0; this.area = 0.0;
}
Light.prototype.calculateArea = function () { this.area = Math.PI * this.radius * this.radius; };
Light.prototype.calculateTBN = function () {
  const n = this.normal;
  const ax = Math.abs(n.x), ay = Math.abs(n.y), az = Math.abs(n.z);
  const minmag = Math.min(ax, ay, az);
  const V = v3(n.x, n.y, n.z);
  if (minmag == ax) V.x = 1.0; else if (minmag == ay) V.y = 1.0; else V.z = 1.0;
  normalize(V);
  this.T = normalize(cross(V, n));
  this.B = normalize(cross(n, this.T));
};
const pad16 = (a) => { const f = new Float32Array(16); f.set(a); return f; };
Light.prototype.toShadowInfo = function () {
  return pad16([this.position.x, this.position.y, this.position.z, this.T.x, this.T.y, this.T.z, this.B.x, this.B.y, this.B.z, this.radius]);
};
Light.prototype.toSceneRenderInfo = function () {
  return pad16([this.position.x, this.position.y, this.position.z, this.normal.x, this.normal.y, this.normal.z,
    this.irradiance.x, this.irradiance.y, this.irradiance.z, this.area]);
};
Light.prototype.toLightRenderInfo = function () {
  return pad16([this.position.x, this.position.y, this.position.z, this.normal.x, this.normal.y, this.normal.z,
    this.irradiance.x, this.irradiance.y, this.irradiance.z, this.radius]);
};

// ---------------------------------------------------------------------------------------------
// Minimal XML reader: elements + text, enough for scenes/*.xml (no attributes, no entities in
// the data; comments, the declaration and a BOM are skipped).
function parseXML(text) {
  if (text.charCodeAt(0) === 0xfeff) text = text.slice(1);
  const root = { name: "#document", children: [], text: "" };
  const stack = [root];
  const re = /<!--[\s\S]*?-->|<\?[\s\S]*?\?>|<!\[CDATA\[([\s\S]*?)\]\]>|<\/\s*([^\s>]+)\s*>|<\s*([^\s>\/]+)[^>]*?(\/?)>|([^<]+)/g;
  let m;
  while ((m = re.exec(text)) !== null) {
    const top = stack[stack.length - 1];
    if (m[1] !== undefined) top.text += m[1];
    else if (m[2] !== undefined) {
      if (stack.length < 2 || top.name !== m[2]) throw new Error("scene XML: unbalanced </" + m[2] + ">");
      stack.pop();
    } else if (m[3] !== undefined) {
      const el = { name: m[3], children: [], text: "" };
      top.children.push(el);
      if (m[4] !== "/") stack.push(el);
    } else if (m[5] !== undefined) top.text += m[5];
  }
  if (stack.length !== 1) throw new Error("scene XML: unclosed <" + stack[stack.length - 1].name + ">");
  return root;
}
function byTag(el, name, out) {  // getElementsByTagName: descendants, document order
  out = out || [];
  for (const c of el.children) { if (c.name === name) out.push(c); byTag(c, name, out); }
  return out;
}
function first(el, name) {
  const r = byTag(el, name);
  if (!r.length) throw new Error("scene XML: <" + el.name + "> has no <" + name + ">");
  return r[0];
}
const xmlNum = (el, name) => Number(first(el, name).text);
const xmlStr = (el, name) => first(el, name).text;
function xmlVec3(el, name) { const e = first(el, name); return v3(xmlNum(e, "x"), xmlNum(e, "y"), xmlNum(e, "z")); }

// ---------------------------------------------------------------------------------------------
// parseMeshJSON (tri/meshDataVersion1.js:12-78) with the gl-matrix 2.2.1 operations it uses.
// gl-matrix stores into Float32Array, so matrices and transformed vectors are fp32-rounded.
function normalFromMat4(a) {  // mat3.normalFromMat4: inverse-transpose of the upper 3x3, via the 4x4 cofactors
  const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7],
    a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
  const b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10, b03 = a01 * a12 - a02 * a11,
    b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12, b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30,
    b08 = a20 * a33 - a23 * a30, b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
  let det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
  if (!det) return null;
  det = 1.0 / det;
  const o = new Float32Array(9);
  o[0] = (a11 * b11 - a12 * b10 + a13 * b09) * det; o[1] = (a12 * b08 - a10 * b11 - a13 * b07) * det; o[2] = (a10 * b10 - a11 * b08 + a13 * b06) * det;
  o[3] = (a02 * b10 - a01 * b11 - a03 * b09) * det; o[4] = (a00 * b11 - a02 * b08 + a03 * b07) * det; o[5] = (a01 * b08 - a00 * b10 - a03 * b06) * det;
  o[6] = (a31 * b05 - a32 * b04 + a33 * b03) * det; o[7] = (a32 * b02 - a30 * b05 - a33 * b01) * det; o[8] = (a30 * b04 - a31 * b02 + a33 * b00) * det;
  return o;
}
function parseMeshJSON(model) {
  const positions = [], normals = [], matIdx = [];
  const b = new Bounds();
  const out3 = new Float32Array(3);  // vec3.create(): results are rounded to fp32 on store
  let nTriangles = 0;
  const nNodes = model.nodes ? model.nodes.length : 1;
  for (let k = 0; k < nNodes; k++) {
    const m = new Float32Array(16);  // mat4.create() + mat4.copy()
    if (model.nodes) m.set(model.nodes[k].modelMatrix); else { m[0] = m[5] = m[10] = m[15] = 1; }
    const nm = normalFromMat4(m);
    const nMeshes = model.nodes ? model.nodes[k].meshIndices.length : model.meshes.length;
    for (let n = 0; n < nMeshes; n++) {
      const mesh = model.meshes[model.nodes ? model.nodes[k].meshIndices[n] : n];
      const P = mesh.vertexPositions, N = mesh.vertexNormals;
      const xf = (x, y, z) => {  // vec3.transformMat4
        out3[0] = m[0] * x + m[4] * y + m[8] * z + m[12];
        out3[1] = m[1] * x + m[5] * y + m[9] * z + m[13];
        out3[2] = m[2] * x + m[6] * y + m[10] * z + m[14];
      };
      for (let i = 0; i < P.length; i += 3) {
        xf(P[i], P[i + 1], P[i + 2]);
        for (let c = 0; c < 3; c++) { if (out3[c] < b.min[c]) b.min[c] = out3[c]; if (out3[c] > b.max[c]) b.max[c] = out3[c]; }
      }
      const nV = mesh.indices ? mesh.indices.length : P.length / 3;
      const nT = nV / 3;
      nTriangles += nT;
      for (let i = 0; i < nT; i++) {
        for (let j = 0; j < 3; j++) {
          let v = i * 3 + j;
          if (mesh.indices) v = mesh.indices[v];
          xf(P[v * 3], P[v * 3 + 1], P[v * 3 + 2]);
          positions.push(out3[0], out3[1], out3[2]);
          const x = N[v * 3], y = N[v * 3 + 1], z = N[v * 3 + 2];  // vec3.transformMat3
          out3[0] = x * nm[0] + y * nm[3] + z * nm[6];
          out3[1] = x * nm[1] + y * nm[4] + z * nm[7];
          out3[2] = x * nm[2] + y * nm[5] + z * nm[8];
          normals.push(out3[0], out3[1], out3[2]);
        }
        matIdx.push(mesh.materialIndex);
      }
    }
  }
  const materials = [];
  (model.materials || []).forEach((mat) => { for (let c = 0; c < 4; c++) materials.push(mat.diffuseReflectance[c]); });
  return { nTriangles: nTriangles, nMaterials: materials.length / 4, materialIndices: matIdx, materials: materials, bounds: b, positions: positions, normals: normals };
}

// ---------------------------------------------------------------------------------------------
// Uniform-grid builders.  The reference bins each primitive's AABB into cells
// floor((min-bmin)/w) .. floor((max-bmin)/w), clamping the LOW index from below only and the HIGH
// index from above only -- so a primitive lying on the max face (low index n > high index n-1) is
// silently dropped (code.js:942-953, 1581-1592, 1669-1680).  Cells are emitted z-major, then y,
// then x, primitives inside a cell in input order, duplicated per cell.  Two counting passes here.
function cellRange(lo, hi, bmin, w, n) {
  const r = new Array(6);
  for (let c = 0; c < 3; c++) {
    let a = Math.floor((lo[c] - bmin[c]) / w[c]);
    let b = Math.floor((hi[c] - bmin[c]) / w[c]);
    if (a < 0) a = 0;
    if (b >= n) b = n - 1;
    r[c] = a; r[3 + c] = b;
  }
  return r;
}
function buildGrid(count, n, bounds, primBox) {
  const bmin = bounds.min;
  const w = [(bounds.max[0] - bmin[0]) / n, (bounds.max[1] - bmin[1]) / n, (bounds.max[2] - bmin[2]) / n];
  const cells = n * n * n;
  const offsets = new Uint32Array(cells + 1);
  const ranges = new Array(count);
  for (let i = 0; i < count; i++) {
    const bx = primBox(i);
    const r = ranges[i] = cellRange(bx[0], bx[1], bmin, w, n);
    for (let z = r[2]; z <= r[5]; z++) for (let y = r[1]; y <= r[4]; y++) for (let x = r[0]; x <= r[3]; x++) offsets[(z * n + y) * n + x + 1]++;
  }
  for (let c = 0; c < cells; c++) offsets[c + 1] += offsets[c];
  const order = new Uint32Array(offsets[cells]);  // order[slot] = primitive index
  const fill = offsets.slice(0, cells);
  for (let i = 0; i < count; i++) {
    const r = ranges[i];
    for (let z = r[2]; z <= r[5]; z++) for (let y = r[1]; y <= r[4]; y++) for (let x = r[0]; x <= r[3]; x++) order[fill[(z * n + y) * n + x]++] = i;
  }
  return { offsets: offsets, order: order };
}

// splitSphereData (code.js:1554-1641): float4 (c, r^2) + material id per slot
function splitSphereData(scene, nSlabs) {
  const S = scene.spheres;
  const g = buildGrid(S.length, nSlabs, scene.sphereBounds, (i) => {
    const s = S[i];
    return [[s.c.x - s.r, s.c.y - s.r, s.c.z - s.r], [s.c.x + s.r, s.c.y + s.r, s.c.z + s.r]];
  });
  const data = new Float32Array(g.order.length * 4), mat = new Uint32Array(g.order.length);
  g.order.forEach((i, k) => { const s = S[i]; data.set([s.c.x, s.c.y, s.c.z, s.r * s.r], 4 * k); mat[k] = s.matId; });
  return { spheres: data, matid: mat, offsets: g.offsets };
}

function triSlots(order, pos9, nor9) {  // 3 x float4 per slot, w = 0 padding (code.js:1010-1034, 1741-1765)
  const pos = new Float32Array(order.length * 12), nor = new Float32Array(order.length * 12);
  order.forEach((i, k) => {
    for (let v = 0; v < 3; v++) for (let c = 0; c < 3; c++) {
      pos[12 * k + 4 * v + c] = pos9[9 * i + 3 * v + c];
      nor[12 * k + 4 * v + c] = nor9[9 * i + 3 * v + c];
    }
  });
  return { pos: pos, nor: nor };
}

// splitTriangleData (code.js:1643-1772)
function splitTriangleData(scene, nSlabs) {
  const T = scene.triangles;
  const pos9 = [], nor9 = [];
  T.forEach((t) => {
    pos9.push(t.p0.x, t.p0.y, t.p0.z, t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z);
    nor9.push(t.n0.x, t.n0.y, t.n0.z, t.n1.x, t.n1.y, t.n1.z, t.n2.x, t.n2.y, t.n2.z);
  });
  const g = buildGrid(T.length, nSlabs, scene.triangleBounds, (i) => triBox(pos9, i));
  const s = triSlots(g.order, pos9, nor9);
  const mat = new Uint32Array(g.order.length);
  g.order.forEach((i, k) => { mat[k] = T[i].matId; });
  return { pos: s.pos, normal: s.nor, matid: mat, offsets: g.offsets };
}
function triBox(p, i) {
  const o = 9 * i;
  const mn = (a, b, c) => Math.min(Math.min(a, b), c), mx = (a, b, c) => Math.max(Math.max(a, b), c);
  return [[mn(p[o], p[o + 3], p[o + 6]), mn(p[o + 1], p[o + 4], p[o + 7]), mn(p[o + 2], p[o + 5], p[o + 8])],
          [mx(p[o], p[o + 3], p[o + 6]), mx(p[o + 1], p[o + 4], p[o + 7]), mx(p[o + 2], p[o + 5], p[o + 8])]];
}

// Mesh (code.js:94-170): the grid is built on the UNtransformed mesh, then positions and bounds are
// normalised / scaled / translated in doubles; fp32 narrowing happens at upload.
function Mesh(jmesh, nslabs, matId, deferGrid) {
  this.bounds = jmesh.bounds;  // shared object, transformed in place like the reference
  this.ntriangles = jmesh.nTriangles;
  this.nslabs = nslabs;
  this.matId = matId;
  this.steps = [];             // the fp64 per-axis steps normalize/scale/translate apply, in order (op 0 sub, 1 mul, 2 add)
  if (deferGrid) {             // device grid build: keep the soup and the UNtransformed bounds the grid is built on
    this.jmesh = jmesh;
    this.gridBounds = new Bounds(jmesh.bounds.min, jmesh.bounds.max);
    this.posData = new Float64Array(0); this.normalData = new Float64Array(0); this.boxSizeData = null;
    return;
  }
  const g = buildGrid(jmesh.nTriangles, nslabs, jmesh.bounds, (i) => triBox(jmesh.positions, i));  // splitMeshData (code.js:899-1041)
  const order = g.order, P = jmesh.positions, N = jmesh.normals;
  this.posData = new Float64Array(order.length * 12);
  this.normalData = new Float64Array(order.length * 12);
  for (let k = 0; k < order.length; k++) for (let v = 0; v < 3; v++) for (let c = 0; c < 3; c++) {
    this.posData[12 * k + 4 * v + c] = P[9 * order[k] + 3 * v + c];
    this.normalData[12 * k + 4 * v + c] = N[9 * order[k] + 3 * v + c];
  }
  this.boxSizeData = g.offsets;
}
Mesh.prototype._each = function (f) {
  const d = this.posData;
  for (let i = 0; i < d.length; i += 4) for (let c = 0; c < 3; c++) d[i + c] = f(d[i + c], c);
  for (let c = 0; c < 3; c++) { this.bounds.min[c] = f(this.bounds.min[c], c); this.bounds.max[c] = f(this.bounds.max[c], c); }
};
Mesh.prototype.normalize = function () {
  const b = this.bounds;
  const ctr = [(b.max[0] + b.min[0]) / 2.0, (b.max[1] + b.min[1]) / 2.0, (b.max[2] + b.min[2]) / 2.0];
  const maxdim = 1.0 / Math.max(Math.max(b.max[0] - b.min[0], b.max[1] - b.min[1]), b.max[2] - b.min[2]);
  this.steps.push({ op: 0, v: ctr }, { op: 1, v: [maxdim, maxdim, maxdim] });
  this._each((v, c) => (v - ctr[c]) * maxdim);
};
Mesh.prototype.scale = function (s) { const k = [s.x, s.y, s.z]; this.steps.push({ op: 1, v: k }); this._each((v, c) => v * k[c]); };
Mesh.prototype.translate = function (t) { const k = [t.x, t.y, t.z]; this.steps.push({ op: 2, v: k }); this._each((v, c) => v + k[c]); };

// ---------------------------------------------------------------------------------------------
// loadScene (code.js:723-897).  `readFile(relPath)` resolves mesh files relative to the scene's
// page directory (the reference fetches "./tri/x.json" relative to index.html).
function loadScene(xmlText, width, height, readFile, opt) {
  const deferGrid = !!(opt && opt.deferGrids);
  const doc = parseXML(xmlText);
  const cam = new Camera();
  const xc = first(doc, "camera");
  cam.lookAt(xmlVec3(xc, "eye"), xmlVec3(xc, "lookAt"), xmlVec3(xc, "vup"), xmlNum(xc, "fov"), width, height);
  const focal = xmlNum(xc, "focal_length"), lensD = xmlNum(xc, "lens_diameter");

  const lights = byTag(doc, "light").map((e) => {
    const l = new Light();
    l.position = xmlVec3(e, "position"); l.normal = xmlVec3(e, "normal"); l.irradiance = xmlVec3(e, "irradiance");
    l.radius = xmlNum(e, "radius");
    l.calculateArea(); l.calculateTBN();
    return l;
  });

  const materials = [], lookup = {};
  byTag(doc, "material").forEach((e, i) => {
    const c = first(e, "color");
    materials.push({ r: xmlNum(c, "r"), g: xmlNum(c, "g"), b: xmlNum(c, "b"), a: xmlNum(c, "a") });
    lookup[xmlStr(e, "id")] = i;
  });

  const sphereBounds = new Bounds();
  const spheres = byTag(doc, "sphere").map((e) => {
    const s = { c: xmlVec3(e, "center"), r: xmlNum(e, "radius"), matId: lookup[xmlStr(e, "matId")] };
    sphereBounds.merge(new Bounds([s.c.x - s.r, s.c.y - s.r, s.c.z - s.r], [s.c.x + s.r, s.c.y + s.r, s.c.z + s.r]));
    return s;
  });

  const triangleBounds = new Bounds();
  const triangles = byTag(doc, "triangle").map((e) => {
    const t = { p0: xmlVec3(e, "p0"), p1: xmlVec3(e, "p1"), p2: xmlVec3(e, "p2"), n0: xmlVec3(e, "n0"), n1: xmlVec3(e, "n1"),
                n2: xmlVec3(e, "n2"), matId: lookup[xmlStr(e, "matId")] };
    const p = [t.p0.x, t.p0.y, t.p0.z, t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z];
    const bx = triBox(p, 0);
    triangleBounds.merge(new Bounds(bx[0], bx[1]));
    return t;
  });
  for (let i = 0; i < 3; i++) {  // flat axis-aligned sets get +-0.1 of thickness (code.js:834-842)
    if (triangleBounds.min[i] == triangleBounds.max[i]) { triangleBounds.min[i] -= 0.1; triangleBounds.max[i] += 0.1; }
  }

  const sceneBounds = new Bounds();
  const meshes = byTag(doc, "mesh").map((e) => {
    // opt.parseMesh: the device ingest of the renderer (queue.meshIngest) in place of the host's parseMeshJSON; it returns the same
    // record with device soups (positionsBuf / normalsBuf) instead of the positions / normals arrays
    const jm = (opt && opt.parseMesh ? opt.parseMesh : parseMeshJSON)(JSON.parse(readFile(xmlStr(e, "file"))));
    const mesh = new Mesh(jm, xmlNum(e, "nslabs"), lookup[xmlStr(e, "matId")], deferGrid);
    if (xmlStr(e, "normalize") == "yes") mesh.normalize();
    mesh.scale(xmlVec3(e, "scale"));
    mesh.translate(xmlVec3(e, "translate"));
    sceneBounds.merge(mesh.bounds);
    return mesh;
  });
  sceneBounds.merge(sphereBounds);
  sceneBounds.merge(triangleBounds);

  return { camera: cam, focal_length: focal, lens_diameter: lensD, lights: lights, materials: materials, bounds: sceneBounds,
           spheres: spheres, sphereBounds: sphereBounds, triangles: triangles, triangleBounds: triangleBounds, meshes: meshes };
}

function loadSceneFile(file, width, height, opt) {
  const pageDir = path.dirname(path.dirname(path.resolve(file)));  // scenes/x.xml -> the page directory
  return loadScene(fs.readFileSync(file, "utf8"), width, height, (rel) => {
    let t = fs.readFileSync(path.resolve(pageDir, rel), "utf8");
    if (t.charCodeAt(0) === 0xfeff) t = t.slice(1);
    return t;
  }, opt);
}

// splitMaterialData (code.js:1774-1782)
function splitMaterialData(scene) {
  const f = new Float32Array(scene.materials.length * 4);
  scene.materials.forEach((c, i) => f.set([c.r, c.g, c.b, c.a], 4 * i));
  return f;
}

// Everything the device needs for one scene, as typed arrays -- and, via toJSON(), the same
// dictionary the test tooling extracts from the reference host (tests compare the two).
function packScene(scene, width, height, raysPerPixel, nSlabs, headerOnly) {
  nSlabs = nSlabs || 1;  // loose spheres / triangles: n_slabs is fixed at 1 in A10 (code.js:399)
  const p = { width: width, height: height, rays_per_pixel: raysPerPixel, n_slabs: nSlabs,
              cam: scene.camera.toFloat32Array(), focal_length: new Float32Array([scene.focal_length])[0],
              lens_rad: new Float32Array([scene.lens_diameter / 2.0])[0], bounds: bounds2AABB(scene.bounds),
              n_spheres: scene.spheres.length, n_triangles: scene.triangles.length };
  if (scene.spheres.length > 0) {
    p.sphere_bounds = bounds2AABB(scene.sphereBounds);
    if (!headerOnly) { const s = splitSphereData(scene, nSlabs); p.spheres = s.spheres; p.s_matid = s.matid; p.s_box = s.offsets; }
  }
  if (scene.triangles.length > 0) {
    p.triangle_bounds = bounds2AABB(scene.triangleBounds);
    if (!headerOnly) { const t = splitTriangleData(scene, nSlabs); p.t_pos = t.pos; p.t_normal = t.normal; p.t_matid = t.matid; p.t_box = t.offsets; }
  }
  p.meshes = scene.meshes.map((m) => headerOnly ? { matid: m.matId, bounds: bounds2AABB(m.bounds), nslabs: m.nslabs, ntriangles: m.ntriangles }
    : ({ pos: new Float32Array(m.posData), normal: new Float32Array(m.normalData), box: m.boxSizeData,
         matid: m.matId, bounds: bounds2AABB(m.bounds), nslabs: m.nslabs, ntriangles: m.ntriangles }));
  p.lights = scene.lights.map((l) => ({ shadow: l.toShadowInfo(), scene: l.toSceneRenderInfo(), light: l.toLightRenderInfo() }));
  p.materials = splitMaterialData(scene);
  return p;
}
function packedToJSON(p) {
  const arr = (a) => Array.prototype.slice.call(a);
  const o = {};
  for (const k of Object.keys(p)) {
    const v = p[k];
    if (ArrayBuffer.isView(v)) o[k] = arr(v);
    else if (Array.isArray(v)) o[k] = v.map((e) => { const q = {}; for (const kk of Object.keys(e)) q[kk] = ArrayBuffer.isView(e[kk]) ? arr(e[kk]) : e[kk]; return q; });
    else o[k] = v;
  }
  return o;
}

module.exports = { Bounds, Camera, Light, Mesh, parseXML, parseMeshJSON, normalFromMat4, loadScene, loadSceneFile, splitSphereData, splitTriangleData,
                   splitMaterialData, bounds2AABB, buildGrid, packScene, packedToJSON };
