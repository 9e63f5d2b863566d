// host/webcl_record.js -- a WebCL 1.0 object model that executes nothing and RECORDS every call made on it.
//
// Purpose: capture the call stream of an UNMODIFIED page script of the reference (Assign10 code.js driven by
// ./harness.js) -- createBuffer sizes, enqueueWriteBuffer payloads, every setArg form, NDRange shapes,
// finish / read / release order -- as data, so that the same stream can be replayed where the reference tree does
// not exist (the GPU box) through the real runtime (./replay.js -> ./webcl.js -> mirt.node -> libmirt.so).
//
// The recorder answers the host's queries with what the real runtime answers, because the host sizes its buffers and
// NDRanges from them (A10 code.js:645-672, 1064-1076): one GPU device, KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE = 64,
// sizeofRay -> 48, sizeofPoi -> 64.  Every answer is written into the trace and ./replay.js checks that the real
// runtime gives the same one.  All other read-backs answer zeros (the reference only displays them).
//
// Trace format (JSON, numbers and kernel names only -- no source text of the page script or of code.cl):
//   { format: 1, meta: {...}, events: [ {op, ...}, ... ] }   + a side file of concatenated payload bytes (`blob: {off, len}`)
// Object ids are small integers handed out in creation order; payloads of <= 256 bytes are inlined as hex.
"use strict";
const crypto = require("crypto");

const C = require("./webcl.js").WebCL;   // the 16 constants (values only; nothing native is touched by reading them)

const STRUCT_ANSWERS = { sizeofRay: 48, sizeofPoi: 64 };   // SURVEY.md section 8: measured on the compiled reference; mirt's k_sizeof* agree
const INLINE_MAX = 256;

function kernelNames(src) {  // `__kernel void NAME(` outside comments
  const code = String(src).replace(/\/\*[\s\S]*?\*\//g, " ").replace(/\/\/[^\n]*/g, " ");
  const out = [], re = /__kernel\s+void\s+([A-Za-z_]\w*)\s*\(/g;
  let m;
  while ((m = re.exec(code))) out.push(m[1]);
  return out;
}

function makeRecordingWebCL(meta) {
  const events = [], blobs = [];
  let blobBytes = 0, nextId = 1;
  const ev = (e) => { events.push(e); return e; };
  const bytesOf = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength);
  const payload = (ta, nbytes) => {
    const b = bytesOf(ta).slice(0, nbytes === undefined ? ta.byteLength : nbytes);
    if (b.length <= INLINE_MAX) return { type: ta.constructor.name, hex: b.toString("hex") };
    const r = { type: ta.constructor.name, blob: { off: blobBytes, len: b.length } };
    blobs.push(Buffer.from(b)); blobBytes += b.length;   // copy: the host may reuse its array
    return r;
  };

  class RBuffer {
    constructor(ctx, flags, bytes) { this.id = nextId++; this.bytes = this.byteLength = bytes; this.answer = null; ev({ op: "createBuffer", id: this.id, ctx: ctx.id, flags: flags, bytes: bytes }); }
    release() { ev({ op: "release", id: this.id, kind: "buffer" }); }
  }
  class RKernel {
    constructor(program, name) {
      if (!program.kernels.includes(name)) throw Object.assign(new Error("no kernel named " + name), { name: "INVALID_KERNEL_NAME" });
      this.id = nextId++; this.name = name; this.args = [];
      ev({ op: "createKernel", id: this.id, program: program.id, name: name });
    }
    setArg(index, value) {
      if (value instanceof RBuffer) { this.args[index] = value; ev({ op: "setArg", kernel: this.id, index: index, buffer: value.id }); }
      else if (ArrayBuffer.isView(value)) { this.args[index] = null; ev(Object.assign({ op: "setArg", kernel: this.id, index: index, bytes: value.byteLength }, payload(value))); }
      else throw new Error(this.name + ".setArg(" + index + "): neither a buffer nor a typed array");
    }
    getWorkGroupInfo(device, what) {
      if (what !== C.KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE) throw new Error("unsupported getWorkGroupInfo query " + what);
      ev({ op: "getWorkGroupInfo", kernel: this.id, what: what, answer: 64 });
      return 64;
    }
    release() { ev({ op: "release", id: this.id, kind: "kernel" }); }
  }
  class RProgram {
    constructor(ctx, src) {
      this.id = nextId++; this.kernels = kernelNames(src); this.status = null;
      ev({ op: "createProgram", id: this.id, ctx: ctx.id, sourceBytes: Buffer.byteLength(String(src)), sourceSha256: crypto.createHash("sha256").update(String(src)).digest("hex"), kernels: this.kernels });
    }
    build() { this.status = 0; ev({ op: "build", program: this.id }); }
    getBuildInfo(device, what) { return what === C.PROGRAM_BUILD_STATUS ? this.status : ""; }
    createKernel(name) { return new RKernel(this, name); }
    release() { ev({ op: "release", id: this.id, kind: "program" }); }
  }
  class RQueue {
    constructor(ctx) { this.id = nextId++; ev({ op: "createCommandQueue", id: this.id, ctx: ctx.id }); }
    enqueueWriteBuffer(buf, blocking, offset, nbytes, ta, events_) {
      ev(Object.assign({ op: "enqueueWriteBuffer", queue: this.id, buffer: buf.id, blocking: !!blocking, offset: offset, bytes: nbytes, waitList: Array.isArray(events_) ? events_.length : null }, payload(ta, nbytes)));
    }
    enqueueReadBuffer(buf, blocking, offset, nbytes, ta, events_) {
      const e = ev({ op: "enqueueReadBuffer", queue: this.id, buffer: buf.id, blocking: !!blocking, offset: offset, bytes: nbytes, type: ta.constructor.name, waitList: Array.isArray(events_) ? events_.length : null });
      const dst = new Uint8Array(ta.buffer, ta.byteOffset, ta.byteLength);
      dst.fill(0, 0, nbytes);
      if (buf.answer !== null && nbytes === 4 && offset === 0) { new DataView(ta.buffer, ta.byteOffset).setUint32(0, buf.answer, true); e.answer = buf.answer; }
    }
    enqueueNDRangeKernel(kernel, dim, offset, globalWS, localWS) {
      ev({ op: "enqueueNDRangeKernel", queue: this.id, kernel: kernel.id, name: kernel.name, dim: dim, offset: offset === null || offset === undefined ? null : Array.from(offset), global: Array.from(globalWS), local: localWS ? Array.from(localWS) : null });
      if (STRUCT_ANSWERS[kernel.name] !== undefined && kernel.args[0]) kernel.args[0].answer = STRUCT_ANSWERS[kernel.name];
    }
    finish() { ev({ op: "finish", queue: this.id }); }
    release() { ev({ op: "release", id: this.id, kind: "queue" }); }
  }
  class RContext {
    constructor(device) { this.id = nextId++; ev({ op: "createContext", id: this.id, device: device ? device.index : null }); }
    createCommandQueue() { return new RQueue(this); }
    createProgram(src) { return new RProgram(this, src); }
    createBuffer(flags, bytes) { return new RBuffer(this, flags, bytes); }
    release() { ev({ op: "release", id: this.id, kind: "context" }); }
  }
  const device = { index: 0, getInfo: (what) => (what === C.DEVICE_TYPE ? C.DEVICE_TYPE_GPU : "call-trace recorder (no device)") };
  const platform = {
    getInfo: (what) => ({ [C.PLATFORM_NAME]: "mirt call-trace recorder", [C.PLATFORM_VENDOR]: "2015-raytracing_amd", [C.PLATFORM_VERSION]: "WebCL 1.0 subset", [C.PLATFORM_PROFILE]: "FULL_PROFILE", [C.PLATFORM_EXTENSIONS]: "" }[what]),
    getDevices: () => [device],
  };
  const webcl = Object.assign({ getPlatforms: () => [platform], createContext: (d) => new RContext(d) }, C);
  return {
    webcl, WebCL: C,
    trace: () => ({ format: 1, meta: Object.assign({ answers: Object.assign({ preferredMultiple: 64 }, STRUCT_ANSWERS), blobBytes: blobBytes }, meta || {}), events: events }),
    blob: () => Buffer.concat(blobs),
  };
}

module.exports = { makeRecordingWebCL, kernelNames };
