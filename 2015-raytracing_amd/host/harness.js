#!/usr/bin/env node
// host/harness.js -- runs an UNMODIFIED page script of the reference (Assign10 code.js with its lib/ and
// tri/ helpers, wherever that tree lives) on the MI355X runtime: a Node `vm` sandbox whose `webcl` /
// `window.WebCL` are ours (./webcl.js) and whose browser globals are file-backed stubs.
//
//   node harness.js <pageDir> <scene.xml> <width> <height> <sqrtRaysPerPixel> <passes> <out.rgba>
//
// <pageDir> is the directory that holds code.js, code.cl, lib/, tri/, scenes/ (e.g. the reference's
// Assign10-Path_Tracing).  Nothing of the reference is bundled here; without that tree this file does nothing.
"use strict";
const fs = require("fs");
const path = require("path");
const vm = require("vm");
const { webcl, WebCL } = require("./webcl.js");
const { parseXML } = require("./scene.js");

function domOf(el) {  // the sliver of DOM loadScene uses: getElementsByTagName + childNodes[0].nodeValue
  const wrap = (e) => ({
    getElementsByTagName: (n) => { const out = []; (function w(x) { for (const c of x.children) { if (c.name === n) out.push(wrap(c)); w(c); } })(e); return out; },
    childNodes: [{ nodeValue: e.text }],
  });
  return wrap(el);
}

function makeSandbox(pageDir, opts) {
  const elems = {
    canvasElement: { getAttribute: (k) => String(opts[k]), getContext: () => ({ createImageData: (w, h) => ({ data: new Uint8ClampedArray(w * h * 4) }), putImageData: (img) => { sandbox.__frame = img.data; } }) },
    ComputeDevices: { selectedIndex: 0, add() {}, options: [] },
    SceneSel: { selectedIndex: 0, add() {}, options: [] },
  };
  function XHR() {
    this.open = (m, url) => { this.url = url; };
    this.overrideMimeType = () => {};
    this.send = () => {
      let t = fs.readFileSync(path.join(pageDir, this.url), "utf8");
      if (t.charCodeAt(0) === 0xfeff) t = t.slice(1);
      this.responseText = t;
      if (/\.xml$/.test(this.url)) this.responseXML = domOf(parseXML(t));
    };
  }
  const sandbox = {
    console: { log() {} }, alert: (m) => { throw new Error("alert: " + m); }, setTimeout() {}, XMLHttpRequest: XHR,
    webcl: opts.webcl || webcl, WebCL: opts.WebCL || WebCL,
    document: { getElementById: (id) => elems[id] || (elems[id] = { value: "", selectedIndex: 0, innerHTML: "", add() {}, options: [] }),
                createElement: () => ({}) },
  };
  sandbox.window = sandbox;
  vm.createContext(sandbox);
  // prepareInitSeeds draws its seeds from Math.random() (A10 code.js:1140-1146); a reproducible run pins the generator (mulberry32)
  if (opts.randomSeed !== undefined) {
    vm.runInContext(`(function () { var a = ${opts.randomSeed >>> 0};
      Math.random = function () { a = (a + 0x6D2B79F5) | 0; var t = Math.imul(a ^ (a >>> 15), 1 | a);
        t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t; return ((t ^ (t >>> 14)) >>> 0) / 4294967296; }; })();`, sandbox);
  }
  for (const f of ["lib/gl-matrix.js", "lib/utilities.js", "tri/meshDataVersion1.js", "mol/pdbParserV1.js", "code.js"]) {
    const p = path.join(pageDir, f);
    if (fs.existsSync(p)) vm.runInContext(fs.readFileSync(p, "utf8"), sandbox, { filename: f });
  }
  return sandbox;
}

function run(pageDir, sceneName, width, height, sqrtRpp, passes, opts) {
  const sb = makeSandbox(pageDir, Object.assign({ width: width, height: height }, opts || {}));
  const js = (s) => vm.runInContext(s, sb);
  // what main() + the page's controls would have set up (A10 code.js:422-460, 530-571)
  js(`findWebCLDevices(); width=${width}; height=${height};
      canvasCtx = document.getElementById("canvasElement").getContext("2d"); imgData = canvasCtx.createImageData(width, height);
      rays_per_pixel=${sqrtRpp * sqrtRpp}; sceneList=[${JSON.stringify(sceneName)}]; updateScene();`);
  if (js("devices.length") === 0) return { devices: 0, frame: null };
  js("preRender();");
  for (let i = 0; i < passes; i++) js("executeRender();");
  js("postRender();");
  return { devices: js("devices.length"), frame: sb.__frame };
}

if (require.main === module) {
  const a = process.argv.slice(2);
  if (a.length < 7) { process.stderr.write("usage: node harness.js <pageDir> <scene.xml> <width> <height> <sqrtRaysPerPixel> <passes> <out.rgba>\n"); process.exit(2); }
  const r = run(a[0], a[1], +a[2], +a[3], +a[4], +a[5]);
  if (!r.frame) { process.stderr.write("no MI355X visible: nothing rendered\n"); process.exit(3); }
  fs.writeFileSync(a[6], Buffer.from(r.frame.buffer, r.frame.byteOffset, r.frame.byteLength));
}
module.exports = { run, makeSandbox };
