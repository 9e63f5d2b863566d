#!/usr/bin/env node
// host/cli.js -- command-line front end of the JavaScript host.
//   node cli.js pack   <scene.xml> <width> <height> <raysPerPixel>                 -> packed kernel inputs as JSON (stdout)
//   node cli.js render <scene.xml> <width> <height> <raysPerPixel> <passes> <out.rgba> [--granular [--graph|--fusion]] [--device-grid] [--no-acu] [--bounces N] [--seeds file.i32] [--gpus N [--force-rccl]]
//                                                                                   -> RGBA8 frame, or a PPM when <out> ends in .ppm (+ <out>.radiance.f32) via the N-API addon
//   node cli.js pack-frame <1|4|7> <mesh.json|mol.pdb|-> <width> <height> [nSlabs]  -> packed inputs of an Assign01/04/07 frame job (stdout)
//   node cli.js frame      <1|4|7> <mesh.json|-> <width> <height> <nSlabs|0> <out.rgba>  -> RGBA8 frame of that job
//   node cli.js ingest <mesh.json> <out-prefix> [--device]                          -> parseMeshJSON's arrays (<out>.pos.f64, .nor.f64, .meta.json) by the host or the device
//   node cli.js devices                                                             -> what webcl.getPlatforms()/getDevices() report
"use strict";
const fs = require("fs");
const path = require("path");
const scene = require("./scene.js");

// <out> ending in .ppm: a binary PPM (P6) any viewer opens; anything else: the raw RGBA8 frame the page would have put on its canvas
function writeFrame(out, px, w, h) {
  const rgba = Buffer.from(px.buffer, px.byteOffset, px.byteLength);
  if (!/\.ppm$/i.test(out)) { fs.writeFileSync(out, rgba); return; }
  const rgb = Buffer.alloc(w * h * 3);
  for (let i = 0, j = 0; i < w * h * 4; i += 4, j += 3) { rgb[j] = rgba[i]; rgb[j + 1] = rgba[i + 1]; rgb[j + 2] = rgba[i + 2]; }
  fs.writeFileSync(out, Buffer.concat([Buffer.from(`P6\n${w} ${h}\n255\n`, "ascii"), rgb]));
}

function usage() {
  process.stderr.write(fs.readFileSync(__filename, "utf8").split("\n").slice(1, 11).join("\n") + "\n");
  process.exit(2);
}

const [cmd, ...rest] = process.argv.slice(2);
if (cmd === "pack") {
  if (rest.length < 4) usage();
  const [file, w, h, rpp] = [rest[0], +rest[1], +rest[2], +rest[3]];
  const sc = scene.loadSceneFile(file, w, h);
  const p = scene.packedToJSON(scene.packScene(sc, w, h, rpp));
  p.scene = path.basename(file);
  process.stdout.write(JSON.stringify(p));
} else if (cmd === "render") {
  if (rest.length < 6) usage();
  const renderer = require("./renderer.js");
  const opt = { granular: rest.includes("--granular"), graph: rest.includes("--graph"), fusion: rest.includes("--fusion"), deviceGrid: rest.includes("--device-grid"), bounces: 5, seeds: null,
                keepAcu: !rest.includes("--no-acu") };   // --no-acu: a one-pass frame without the 16 bytes per ray: the pass resolves its own pixels (mirt.h)
  let i;
  if ((i = rest.indexOf("--bounces")) >= 0) opt.bounces = +rest[i + 1];
  if ((i = rest.indexOf("--gpus")) >= 0) { opt.gpus = +rest[i + 1]; opt.forceRccl = rest.includes("--force-rccl"); }   // row tiles over N devices + gather
  if ((i = rest.indexOf("--seeds")) >= 0) { const b = fs.readFileSync(rest[i + 1]); opt.seeds = new Int32Array(b.buffer, b.byteOffset, b.length / 4); }
  const [file, w, h, rpp, passes, out] = [rest[0], +rest[1], +rest[2], +rest[3], +rest[4], rest[5]];
  const res = renderer.renderFile(file, w, h, rpp, passes, opt);
  writeFrame(out, res.pixel, w, h);
  fs.writeFileSync(out + ".radiance.f32", Buffer.from(res.radiance.buffer, res.radiance.byteOffset, res.radiance.byteLength));
  process.stderr.write(`rendered ${file} ${w}x${h} rpp ${rpp}, ${passes} pass(es), ${opt.granular ? (opt.fusion ? "kernel-by-kernel, passes fused by the runtime" : "kernel-by-kernel") : "fused"}: ${res.ms.toFixed(2)} ms on ${res.device}; passes the runtime fused from enqueues: ${res.fusedPasses}\n`);
  if (res.routes) process.stderr.write(`gather routes per tile: ${res.routes.join(" ")}; peer access root<-tile: ${res.peerAccess.join(" ")}\n`);
} else if (cmd === "pack-frame" || cmd === "frame") {
  if (rest.length < 4) usage();
  const frame = require("./frame.js");
  const assign = +rest[0], text = rest[1] === "-" ? null : fs.readFileSync(rest[1], "utf8").replace(/^\ufeff/, "");
  const model = text === null ? null : /\.pdb$/i.test(rest[1]) ? { pdb: text } : JSON.parse(text);   // .pdb: Assign07's molecule mode
  const p = frame.packFrame(assign, model, +rest[2], +rest[3], +rest[4] || 2);
  if (cmd === "pack-frame") process.stdout.write(JSON.stringify(scene.packedToJSON(p)));
  else writeFrame(rest[5], frame.renderFrame(p), +rest[2], +rest[3]);
} else if (cmd === "ingest") {
  if (rest.length < 2) usage();
  const model = JSON.parse(fs.readFileSync(rest[0], "utf8").replace(/^\ufeff/, ""));
  let pos, nor, meta;
  if (rest.includes("--device")) {
    const { webcl } = require("./webcl.js");
    const ctx = webcl.createContext(webcl.getPlatforms()[0].getDevices(webcl.DEVICE_TYPE_ALL)[0]), q = ctx.createCommandQueue();
    const r = q.meshIngest(model, scene.normalFromMat4);
    pos = new Float64Array(r.nTriangles * 9); nor = new Float64Array(r.nTriangles * 9);
    if (r.nTriangles) { q.enqueueReadBuffer(r.positionsBuf, true, 0, pos.byteLength, pos, []); q.enqueueReadBuffer(r.normalsBuf, true, 0, nor.byteLength, nor, []); }
    meta = { nTriangles: r.nTriangles, bounds: r.bounds6, materialIndices: r.materialIndices, materials: r.materials };
    ctx.release();
  } else {
    const r = scene.parseMeshJSON(model);
    pos = new Float64Array(r.positions); nor = new Float64Array(r.normals);
    meta = { nTriangles: r.nTriangles, bounds: r.bounds.min.concat(r.bounds.max), materialIndices: r.materialIndices, materials: r.materials };
  }
  fs.writeFileSync(rest[1] + ".pos.f64", Buffer.from(pos.buffer));
  fs.writeFileSync(rest[1] + ".nor.f64", Buffer.from(nor.buffer));
  fs.writeFileSync(rest[1] + ".meta.json", JSON.stringify(meta));
} else if (cmd === "devices") {
  const { webcl } = require("./webcl.js");
  for (const p of webcl.getPlatforms()) {
    console.log(p.getInfo(webcl.PLATFORM_NAME), "|", p.getInfo(webcl.PLATFORM_VENDOR), "|", p.getInfo(webcl.PLATFORM_VERSION));
    for (const d of p.getDevices(webcl.DEVICE_TYPE_ALL)) console.log("  device:", d.getInfo(webcl.DEVICE_NAME));
  }
} else usage();
