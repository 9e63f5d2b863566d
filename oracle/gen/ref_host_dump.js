// oracle/gen/ref_host_dump.js -- TEST INFRASTRUCTURE, build-container only.
//
// Runs the REFERENCE's own host code (Assign10 code.js + lib/ + tri/ parser,
// read from /root/reference at run time, never copied) inside a Node `vm`
// sandbox with file-backed XMLHttpRequest / DOM stubs, and dumps every input
// buffer the reference would hand to its kernels for one scene, as JSON on
// stdout.  tests/golden/ fixtures are generated from this (gen_golden.py); our
// own JS host (2015-raytracing_amd/host/) is tested against the same dump.
//
// usage: node ref_host_dump.js <REFROOT> <scene.xml> <width> <height> <rays_per_pixel> [n_slabs] [pageDir]
//   pageDir: where scenes/ and tri/ are read from (default: the reference's own Assign10 page); the host CODE
//   always comes from REFROOT.
"use strict";
const fs = require("fs");
const path = require("path");
const vm = require("vm");
const { JSDOM } = require("/usr/share/nodejs/jsdom");

const [refroot, sceneName, W, H, RPP, NSLABS, PAGEDIR] = process.argv.slice(2);
const adir = path.join(refroot, "Assign10-Path_Tracing");
const datadir = PAGEDIR ? path.resolve(PAGEDIR) : adir;

function XHR() {
  this.open = function (m, url) { this.url = url; };
  this.overrideMimeType = function () {};
  this.send = function () {
    let txt = fs.readFileSync(path.join(datadir, this.url), "utf8");
    if (txt.charCodeAt(0) === 0xfeff) txt = txt.slice(1);
    this.responseText = txt;
    if (/\.xml$/.test(this.url)) {
      this.responseXML = new JSDOM(txt, { contentType: "text/xml" }).window.document;
    }
  };
}
const elems = {};
const sandbox = {
  console: { log: function () {} },
  alert: function (m) { throw new Error("alert: " + m); },
  setTimeout: function () {},
  XMLHttpRequest: XHR,
  document: {
    getElementById: function (id) {
      if (!elems[id]) elems[id] = { value: "", selectedIndex: 0, innerHTML: "" };
      return elems[id];
    },
  },
};
sandbox.window = sandbox; // gl-matrix attaches to window
vm.createContext(sandbox);
for (const f of ["lib/gl-matrix.js", "lib/utilities.js", "tri/meshDataVersion1.js", "mol/pdbParserV1.js", "code.js"]) {
  vm.runInContext(fs.readFileSync(path.join(adir, f), "utf8"), sandbox, { filename: f });
}
// the globals the page's controls would set (code.js:396-402, 444-446, 530-543)
vm.runInContext(`width=${+W}; height=${+H}; rays_per_pixel=${+RPP}; n_slabs=${+NSLABS > 0 ? +NSLABS : 1};`, sandbox);

const out = vm.runInContext(`(function(){
  var scene = loadScene("scenes/${sceneName}");
  var f32 = function(a){ return Array.prototype.slice.call(new Float32Array(a)); };
  var u32 = function(a){ return Array.prototype.slice.call(new Uint32Array(a)); };
  var o = { scene: "${sceneName}", width: width, height: height, rays_per_pixel: rays_per_pixel, n_slabs: n_slabs };
  o.cam = f32(scene.camera.toFloat32Array());
  o.focal_length = f32([scene.focal_length])[0];
  o.lens_rad = f32([scene.lens_diameter/2.0])[0];
  o.bounds = f32(bounds2AABB(scene.bounds));
  o.n_spheres = scene.spheres.length; o.n_triangles = scene.triangles.length;
  if (scene.spheres.length > 0) {
    var sd=[], sm=[], sb=[]; splitSphereData(scene, sd, sm, sb);
    o.spheres = f32(sd); o.s_matid = u32(sm); o.s_box = u32(sb); o.sphere_bounds = f32(bounds2AABB(scene.sphereBounds));
  }
  if (scene.triangles.length > 0) {
    var pd=[], nd=[], md=[], tb=[]; splitTriangleData(scene, pd, nd, md, tb);
    o.t_pos = f32(pd); o.t_normal = f32(nd); o.t_matid = u32(md); o.t_box = u32(tb); o.triangle_bounds = f32(bounds2AABB(scene.triangleBounds));
  }
  o.meshes = [];
  for (var i = 0; i < scene.meshes.length; i++) {
    var m = scene.meshes[i];
    o.meshes.push({ pos: f32(m.posData), normal: f32(m.normalData), box: u32(m.boxSizeData),
                    matid: m.matId, bounds: f32(bounds2AABB(m.bounds)), nslabs: m.nslabs, ntriangles: m.ntriangles });
  }
  o.lights = [];
  for (var i = 0; i < scene.lights.length; i++) {
    var l = scene.lights[i];
    o.lights.push({ shadow: f32(l.toShadowInfo()), scene: f32(l.toSceneRenderInfo()), light: f32(l.toLightRenderInfo()) });
  }
  var md2 = []; splitMaterialData(scene, md2); o.materials = f32(md2);
  return o;
})()`, sandbox);
process.stdout.write(JSON.stringify(out));
