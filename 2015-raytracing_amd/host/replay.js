#!/usr/bin/env node
// host/replay.js -- replays a recorded WebCL call trace (./webcl_record.js) through the REAL runtime:
// every event becomes the same call on ./webcl.js (-> mirt.node -> libmirt.so -> HIP kernels), in the recorded order,
// with the recorded payloads, argument forms, NDRange shapes, finish() points and release order.
//
// The traces under tests/golden/calltrace_* were recorded from the reference's unmodified Assign10 code.js
// (preRender -> executeRender x passes -> postRender, A10 code.js:1784-1859); replaying one is running that host's
// call stream on an MI355X without needing the reference tree there.
//
//   node replay.js <trace-prefix> <out-prefix>
//     <out-prefix>.reads.bin   the bytes of every enqueueReadBuffer, concatenated in order
//     <out-prefix>.acu.bin / .seeds.bin   final contents of the accumulator (arg 0 of initAcu) and the seed buffer
//                                         (arg 0 of initTrace), read just before the trace releases them
//     stdout                   JSON report: events replayed, answers checked, device name
"use strict";
const fs = require("fs");
const zlib = require("zlib");
const real = require("./webcl.js");

const TYPES = { Int8Array, Uint8Array, Uint8ClampedArray, Int16Array, Uint16Array, Int32Array, Uint32Array, Float32Array, Float64Array };

function replay(trace, blob, hooks) {
  hooks = hooks || {};
  const webcl = hooks.webcl || real.webcl;   // tests replay onto a second recorder to prove the replayer re-issues the stream unchanged
  const obj = new Map(), roles = new Map(), reads = [], dumps = {}, checked = { preferredMultiple: 0, structSizes: 0 };
  let fusedPasses = 0;
  const get = (id) => { const o = obj.get(id); if (!o) throw new Error("trace refers to unknown object " + id); return o; };
  const typed = (e, bytes) => { const T = TYPES[e.type]; const c = Buffer.from(bytes); return new T(c.buffer, c.byteOffset, c.length / T.BYTES_PER_ELEMENT); };
  const payload = (e) => typed(e, e.hex !== undefined ? Buffer.from(e.hex, "hex") : blob.slice(e.blob.off, e.blob.off + e.blob.len));
  const devices = webcl.getPlatforms()[0].getDevices(webcl.DEVICE_TYPE_ALL);
  if (!devices.length) throw new Error("no MI355X visible: a trace can only be replayed on the device (no CPU path)");
  let queue = null;
  for (const e of trace.events) {
    switch (e.op) {
      case "createContext": obj.set(e.id, webcl.createContext(devices[e.device || 0])); break;
      case "createCommandQueue": queue = get(e.ctx).createCommandQueue(); obj.set(e.id, queue); break;
      case "createProgram":   // the trace carries kernel names, not source text: hand the runtime the manifest form of the same program
        obj.set(e.id, get(e.ctx).createProgram(e.kernels.map((k) => "__kernel void " + k + "();").join("\n"))); break;
      case "build": get(e.program).build(); break;
      case "createKernel": { const k = get(e.program).createKernel(e.name); k.traceName = e.name; obj.set(e.id, k); break; }
      case "createBuffer": obj.set(e.id, get(e.ctx).createBuffer(e.flags, e.bytes)); break;
      case "setArg":
        if (e.buffer !== undefined) {
          const k = get(e.kernel);
          k.setArg(e.index, get(e.buffer));
          if (e.index === 0 && k.traceName === "initAcu") roles.set(e.buffer, "acu");
          if (e.index === 0 && k.traceName === "initTrace") roles.set(e.buffer, "seeds");
        } else get(e.kernel).setArg(e.index, payload(e));
        break;
      case "getWorkGroupInfo": {
        const a = get(e.kernel).getWorkGroupInfo(devices[0], e.what);
        if (a !== e.answer) throw new Error(`getWorkGroupInfo answered ${a}, the trace was recorded with ${e.answer}: its NDRange shapes do not apply`);
        checked.preferredMultiple++;
        break;
      }
      case "enqueueWriteBuffer": get(e.queue).enqueueWriteBuffer(get(e.buffer), e.blocking, e.offset, e.bytes, payload(e), []); break;
      case "enqueueNDRangeKernel": get(e.queue).enqueueNDRangeKernel(get(e.kernel), e.dim, e.offset, e.global, e.local); break;
      case "enqueueReadBuffer": {
        const T = TYPES[e.type], dst = new T(new ArrayBuffer(Math.ceil(e.bytes / T.BYTES_PER_ELEMENT) * T.BYTES_PER_ELEMENT));
        get(e.queue).enqueueReadBuffer(get(e.buffer), e.blocking, e.offset, e.bytes, dst, []);
        reads.push({ event: e, dst: dst });
        break;
      }
      case "finish":
        get(e.queue).finish();
        for (const r of reads) if (r.event.answer !== undefined && !r.checked) {   // non-blocking reads are complete after finish()
          const got = new DataView(r.dst.buffer).getUint32(0, true);
          if (got !== r.event.answer) throw new Error(`struct size read back as ${got}, the trace was recorded with ${r.event.answer}: its buffer sizes do not apply`);
          r.checked = true; checked.structSizes++;
        }
        break;
      case "release": {
        const o = get(e.id);
        if (e.kind === "buffer" && roles.has(e.id) && !hooks.noDumps) {   // extra read-back of ours, outside the recorded stream
          const out = new Uint8Array(o.byteLength);
          queue.enqueueReadBuffer(o, true, 0, o.byteLength, out, []); queue.finish();
          dumps[roles.get(e.id)] = out;
        }
        if (e.kind === "context" && typeof o.fusedPasses === "function") fusedPasses += o.fusedPasses();   // ours: how many passes ran fused (MIRT_FUSION=2)
        o.release(); obj.delete(e.id);
        break;
      }
      default: throw new Error("unknown trace event " + e.op);
    }
    if (hooks.after) hooks.after(e);
  }
  return { reads: reads.map((r) => Buffer.from(r.dst.buffer, 0, r.event.bytes)), dumps: dumps, checked: checked, device: devices[0].getInfo(webcl.DEVICE_NAME), leaked: obj.size, fusedPasses: fusedPasses };
}

function load(prefix) {
  const trace = JSON.parse(fs.readFileSync(prefix + ".json", "utf8"));
  const blob = zlib.gunzipSync(fs.readFileSync(prefix + ".bin.gz"));
  if (blob.length !== trace.meta.blobBytes) throw new Error("payload side file does not match the trace");
  return { trace, blob };
}

if (require.main === module) {
  const a = process.argv.slice(2);
  if (a.length < 2) { process.stderr.write("usage: node replay.js <trace-prefix> <out-prefix>\n"); process.exit(2); }
  const { trace, blob } = load(a[0]);
  const t0 = process.hrtime.bigint();
  const r = replay(trace, blob);
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  fs.writeFileSync(a[1] + ".reads.bin", Buffer.concat(r.reads));
  for (const k of Object.keys(r.dumps)) fs.writeFileSync(a[1] + "." + k + ".bin", Buffer.from(r.dumps[k].buffer));
  process.stdout.write(JSON.stringify({ events: trace.events.length, reads: r.reads.map((b) => b.length), checked: r.checked, device: r.device, leaked: r.leaked, fusedPasses: r.fusedPasses, ms: ms }) + "\n");
}
module.exports = { replay, load };
