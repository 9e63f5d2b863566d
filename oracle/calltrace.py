"""oracle/calltrace.py -- TEST INFRASTRUCTURE (checker only).

Plays a recorded WebCL call trace (tests/golden/calltrace_*.json + .bin.gz, made by oracle/gen/record_calltrace.js
from the reference's UNMODIFIED Assign10 code.js) against a CPU implementation of the fourteen kernels:

  * oracle/_ref/libref_a10.so   the reference's own code.cl compiled for x86 (build container only) -> the
                                expected read-backs committed next to the trace (oracle/gen/gen_calltrace.py)
  * oracle/liboracle.so         our C restatement (travels to the GPU box; used to localise a mismatch)

The player is a tiny WebCL "device": buffers are numpy byte arrays, setArg stores bytes or buffer ids, an
enqueueNDRangeKernel calls the CPU kernel with the arguments bound at that moment and the global size of the event
(the CPU kernels walk work-items 0..global-1 row-major, the pinned order of the seeds race),
enqueueReadBuffer snapshots the bytes.  Argument kinds per kernel follow A10 code.cl:440-1386.
"""
import gzip
import json
import os

import numpy as np

# argument kinds in the kernels' own order (A10 code.cl): B buffer, U uint, F float, V float16, X AABB (8 floats: min,1,max,1)
KERNEL_ARGS = {
    "sizeofRay": "B", "sizeofPoi": "B",
    "initAcu": "BU",
    "initTrace": "BBBXVFFU",
    "sphereTrace": "UBBBBBXU",
    "triangleTrace": "UBBBBBBXU",
    "meshTrace": "UBBBBBUXU",
    "lightRender": "BBBVU",
    "initShadowTrace": "BBUVB",
    "sphereShadowTrace": "UBBBXU",
    "triangleShadowTrace": "UBBBXU",
    "sceneRender": "BBBBVU",
    "bouncePaths": "BBBU",
    "copyToPixel": "BBFUU",
}
SCALAR_BYTES = {"U": 4, "F": 4, "V": 64, "X": 32}


def load_trace(prefix):
    """-> (trace dict, payload bytes).  `prefix` = path without .json / .bin.gz"""
    with open(prefix + ".json") as f:
        t = json.load(f)
    with gzip.open(prefix + ".bin.gz", "rb") as f:
        blob = f.read()
    assert len(blob) == t["meta"]["blobBytes"], "payload side file does not match the trace"
    return t, blob


def payload(e, blob):
    if "hex" in e:
        return bytes.fromhex(e["hex"])
    b = e["blob"]
    return blob[b["off"]:b["off"] + b["len"]]


def play(trace, blob, k):
    """Run the trace on CpuKernels `k` (oracle/a10_pass.py).  Returns {"reads": [bytes per enqueueReadBuffer, in order],
    "buffers": {name: np.uint8 array}} where buffers holds the final contents of `acu` (arg 0 of initAcu), `seeds`
    (arg 0 of initTrace) and `pixel` (arg 0 of copyToPixel), captured just before they are released."""
    import ctypes as C
    bufs, kern, reads, named, keep = {}, {}, [], {}, {}
    roles = {}
    gpu = getattr(k, "on_gpu", False)   # oracle/ref_gpu.GpuRefKernels: buffer arguments are device mirrors of the host arrays (buf / upload / flush)
    for e in trace["events"]:
        op = e["op"]
        if op == "createBuffer":
            bufs[e["id"]] = np.zeros(max(e["bytes"], 4), np.uint8)
        elif op == "createKernel":
            kern[e["id"]] = {"name": e["name"], "args": {}}
        elif op == "setArg":
            kern[e["kernel"]]["args"][e["index"]] = ("B", e["buffer"]) if "buffer" in e else ("S", payload(e, blob))
            name = kern[e["kernel"]]["name"]
            if "buffer" in e and (name, e["index"]) in (("initAcu", 0), ("initTrace", 0), ("copyToPixel", 0)):
                roles[e["buffer"]] = {"initAcu": "acu", "initTrace": "seeds", "copyToPixel": "pixel"}[name]
        elif op == "enqueueWriteBuffer":
            data = np.frombuffer(payload(e, blob), np.uint8)
            assert len(data) == e["bytes"]
            bufs[e["buffer"]][e["offset"]:e["offset"] + e["bytes"]] = data
            if gpu:
                k.upload(bufs[e["buffer"]])
        elif op == "enqueueReadBuffer":
            if gpu:
                k.flush()
            reads.append(bufs[e["buffer"]][e["offset"]:e["offset"] + e["bytes"]].tobytes())
        elif op == "enqueueNDRangeKernel":
            kk = kern[e["kernel"]]
            name, kinds = kk["name"], KERNEL_ARGS[kk["name"]]
            assert e["offset"] is None
            if name in ("sizeofRay", "sizeofPoi"):
                bufs[kk["args"][0][1]][:4] = np.frombuffer(np.uint32(getattr(k, name)()).tobytes(), np.uint8)
                continue
            args = []
            for i, kind in enumerate(kinds):
                assert i in kk["args"], f"{name}: argument {i} was never set"
                tag, v = kk["args"][i]
                if kind == "B":
                    assert tag == "B", f"{name} arg {i}: expected a buffer"
                    args.append(k.buf(bufs[v]) if gpu else bufs[v].ctypes.data_as(C.c_void_p))
                else:
                    assert tag == "S" and len(v) == SCALAR_BYTES[kind], f"{name} arg {i}: {len(v)} bytes for kind {kind}"
                    if kind == "U":
                        args.append(int(np.frombuffer(v, np.uint32)[0]))
                    elif kind == "F":
                        args.append(float(np.frombuffer(v, np.float32)[0]))
                    else:
                        a = np.frombuffer(v, np.float32).copy()
                        keep[(e["kernel"], i)] = a
                        args.append(a.ctypes.data_as(C.POINTER(C.c_float)))
            assert e["dim"] == len(e["global"]) == (2 if name == "initTrace" else 1)
            getattr(k, name)(*args, *e["global"])
        elif op == "release" and e["kind"] == "buffer":
            if gpu:
                k.flush()
            if e["id"] in roles:
                named[roles[e["id"]]] = bufs[e["id"]]
            else:
                del bufs[e["id"]]
    if gpu:
        k.flush()
    for i, r in roles.items():   # traces that never release
        named.setdefault(r, bufs.get(i))
    return {"reads": reads, "buffers": named}


def launch_shapes(trace):
    """[(kernel name, dim, global, local)] of every enqueueNDRangeKernel, in order."""
    return [(e["name"], e["dim"], tuple(e["global"]), tuple(e["local"]) if e["local"] else None)
            for e in trace["events"] if e["op"] == "enqueueNDRangeKernel"]


def golden_prefix(name):
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "calltrace_" + name)
