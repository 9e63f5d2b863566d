#!/usr/bin/env node
// oracle/gen/record_calltrace.js -- TEST INFRASTRUCTURE, build container only.
//
// Runs the reference's UNMODIFIED Assign10 page script (code.js + lib/ + tri/, read where they lie under
// <refroot>) in host/harness.js' sandbox against the recording WebCL of host/webcl_record.js, through
//     findWebCLDevices(); updateScene(); preRender(); executeRender() x passes; postRender()
// and writes the call stream it issued:
//     <out>.json       events (numbers and kernel names only)
//     <out>.bin.gz     payload bytes of the enqueueWriteBuffer / setArg calls above 256 bytes
// Math.random is pinned (mulberry32(randomSeed)) so prepareInitSeeds (code.js:1140-1146) is reproducible.
//
//   node record_calltrace.js <refroot> <scene.xml> <width> <height> <sqrtRaysPerPixel> <passes> <randomSeed> <out-prefix> [pageDirOverride]
"use strict";
const fs = require("fs");
const path = require("path");
const zlib = require("zlib");
const HOST = path.join(__dirname, "..", "..", "2015-raytracing_amd", "host");
const { run } = require(path.join(HOST, "harness.js"));
const { makeRecordingWebCL } = require(path.join(HOST, "webcl_record.js"));

const a = process.argv.slice(2);
if (a.length < 8) { process.stderr.write("usage: see the header of this file\n"); process.exit(2); }
const [refroot, scene, width, height, k, passes, rseed, out] = [a[0], a[1], +a[2], +a[3], +a[4], +a[5], +a[6], a[7]];
const page = a[8] || path.join(refroot, "Assign10-Path_Tracing");
const rec = makeRecordingWebCL({ page: "Assign10-Path_Tracing", scene: scene, width: width, height: height, sqrtRaysPerPixel: k,
                                 raysPerPixel: k * k, passes: passes, mathRandom: "mulberry32", randomSeed: rseed,
                                 driver: "findWebCLDevices; updateScene; preRender; executeRender x passes; postRender" });
const r = run(page, scene, width, height, k, passes, { webcl: rec.webcl, WebCL: rec.WebCL, randomSeed: rseed });
if (!r.devices) throw new Error("the page script found no device on the recorder");
const t = rec.trace();
fs.writeFileSync(out + ".json", JSON.stringify(t));
fs.writeFileSync(out + ".bin.gz", zlib.gzipSync(rec.blob(), { level: 9 }));
const n = {};
for (const e of t.events) n[e.op] = (n[e.op] || 0) + 1;
process.stderr.write(`${scene} ${width}x${height} k=${k} passes=${passes}: ${t.events.length} events ${JSON.stringify(n)}, ${t.meta.blobBytes} payload bytes\n`);
