// oracle/gen/ref_host_dump_frame.js -- TEST INFRASTRUCTURE, build-container only.
//
// Runs the REFERENCE's own Assign01 / Assign04 / Assign07 host code (code.js + lib/ + tri/ parser, read from
// /root/reference at run time, never copied) in a Node `vm` sandbox and dumps the buffers computeTri()/compute()
// would upload for one mesh: camera pack, AABB, float4-padded positions / normals, material indices, colours,
// and (A07) the grid-sorted arrays with their cell offsets.
//
// usage: node ref_host_dump_frame.js <REFROOT> <1|4|7> <mesh.json|-> <width> <height> [n_slabs] [pageDir]
"use strict";
const fs = require("fs");
const path = require("path");
const vm = require("vm");

const [refroot, assign, meshName, W, H, NSLABS, PAGEDIR] = process.argv.slice(2);
const dirs = { 1: "Assign01-Sphere_Ray_Tracing", 4: "Assign04-Triangle_Mesh", 7: "Assign07-3D_uniform_grid_acceleration" };
const adir = path.join(refroot, dirs[assign]);
const datadir = PAGEDIR ? path.resolve(PAGEDIR) : adir;

function XHR() {
  this.open = function (m, url) { this.url = url; };
  this.overrideMimeType = function () {};
  this.send = function () {
    let txt = fs.readFileSync(path.join(datadir, this.url), "utf8");
    if (txt.charCodeAt(0) === 0xfeff) txt = txt.slice(1);
    this.responseText = txt;
  };
}
const elems = {};
const sandbox = { console: { log: function () {} }, alert: function (m) { throw new Error("alert: " + m); }, setTimeout: function () {},
  XMLHttpRequest: XHR, document: { getElementById: function (id) { return elems[id] || (elems[id] = { value: "", selectedIndex: 0 }); } } };
sandbox.window = sandbox;
vm.createContext(sandbox);
for (const f of ["lib/gl-matrix.js", "lib/utilities.js", "tri/meshDataVersion1.js", "mol/pdbParserV1.js", "code.js"]) {
  const p = path.join(adir, f);
  if (fs.existsSync(p)) vm.runInContext(fs.readFileSync(p, "utf8"), sandbox, { filename: f });
}
const f32 = "function(a){ return Array.prototype.slice.call(new Float32Array(a)); }";
const u32 = "function(a){ return Array.prototype.slice.call(new Uint32Array(a)); }";
let out;
if (assign === "1") {
  out = vm.runInContext(`(function(){ var f32=${f32}; var cam = new Camera(); cam.defaultInit(); cam.width = 2.66; cam.height = 2.0; cam.cols = ${+W}; cam.rows = ${+H};
    return { assign: 1, width: ${+W}, height: ${+H}, cam: f32(cam.toFloat32Array()) }; })()`, sandbox);   // A01 code.js:180-185
} else if (/\.pdb$/.test(meshName)) {
  // A07 molecule mode: compute() (A07 code.js:569-600): parsePDB, cam.set(bounds), splitMolData; `pdb` keeps parsePDB's own output
  vm.runInContext(`width=${+W}; height=${+H}; n_slabs=${+NSLABS > 0 ? +NSLABS : 2};`, sandbox);
  out = vm.runInContext(`(function(){
    var f32=${f32}, u32=${u32};
    var molData = parsePDB(loadFromFile("mol/${meshName}"));
    cam.defaultInit(); cam.set(molData.bounds, width, height);
    var pd=[], id=[], sd=[]; splitMolData(molData, pd, id, sd);
    return { assign: 7, mol: "${meshName}", width: width, height: height, cam: f32(cam.toFloat32Array()), n_slabs: n_slabs, s_size: molData.size,
             bounds: f32([molData.bounds.min[0], molData.bounds.min[1], molData.bounds.min[2], 1, molData.bounds.max[0], molData.bounds.max[1], molData.bounds.max[2], 1]),
             atoms: f32(pd), mindex: u32(id), slab_size: u32(sd), mcolor: f32(molData.colorData),
             pdb: { size: molData.size, atomData: molData.atomData, colorData: molData.colorData, radiusData: molData.radiusData,
                    min: molData.bounds.min, max: molData.bounds.max } }; })()`, sandbox);
} else {
  vm.runInContext(`width=${+W}; height=${+H}; ${assign === "7" ? "n_slabs=" + (+NSLABS > 0 ? +NSLABS : 2) + ";" : ""}`, sandbox);
  out = vm.runInContext(`(function(){
    var f32=${f32}, u32=${u32};
    var meshData = parseMeshJSON("tri/${meshName}");
    cam.defaultInit(); cam.set(meshData.bounds, width, height);           // computeTri (A04 code.js:553-577, A07 code.js:603-628)
    var o = { assign: ${+assign}, mesh: "${meshName}", width: width, height: height, cam: f32(cam.toFloat32Array()),
              bounds: f32([meshData.bounds.min[0], meshData.bounds.min[1], meshData.bounds.min[2], 1, meshData.bounds.max[0], meshData.bounds.max[1], meshData.bounds.max[2], 1]), t_size: meshData.nTriangles, mcolor: f32(meshData.materials) };
    if (${+assign} === 4) {
      o.pos = f32(toPosArray(meshData)); o.normal = f32(toNormalArray(meshData)); o.mindex = u32(meshData.materialIndices);
    } else {
      var pd=[], nd=[], id=[], sd=[]; splitMeshData(meshData, pd, nd, id, sd);
      o.pos = f32(pd); o.normal = f32(nd); o.mindex = u32(id); o.slab_size = u32(sd); o.n_slabs = n_slabs;
    }
    return o; })()`, sandbox);
}
process.stdout.write(JSON.stringify(out));
