#!/usr/bin/env python3
"""oracle/gen/gen_calltrace.py -- TEST INFRASTRUCTURE, build container only.

For each case: (1) record the call stream of the reference's UNMODIFIED Assign10 code.js on the recording WebCL
(record_calltrace.js; Math.random pinned) -> tests/golden/calltrace_<name>.json + .bin.gz; (2) play that stream
against the reference's own code.cl compiled for x86 (oracle/_ref, oracle/calltrace.py) -> what every
enqueueReadBuffer of the stream returns, plus the final per-pixel radiance sums (at one ray per pixel: the accumulator
itself) and SHA-256 digests of the final accumulator and seed buffers -> tests/golden/calltrace_<name>_expect.npz.  The fixtures hold numbers and kernel names only.

    make -C oracle ref && python oracle/gen/gen_calltrace.py
"""
import hashlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import a10_pass as A  # noqa: E402
import calltrace as CT  # noqa: E402

REFROOT = os.environ.get("REFROOT", "/root/reference")
# (name, scene, width, height, sqrt(rays per pixel), passes, Math.random seed): the page's own 320x240 canvas (A10 index.html:46)
CASES = [
    ("cornell_320x240_k1_p2", "cornell.xml", 320, 240, 1, 2, 20150410),
    ("cornell_320x240_k2_p2", "cornell.xml", 320, 240, 2, 2, 20150411),
    ("cornell_teapot3_320x240_k1_p2", "cornell_teapot3.xml", 320, 240, 1, 2, 20150412),
    ("threeLights_160x120_k3_p3", "threeLights.xml", 160, 120, 3, 3, 20150413),   # odd lens grid (NaN sample), 3 lights, 3 passes
]


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def main():
    ref = A.load_ref()
    only = set(sys.argv[1:])
    for name, scene, w, h, k, passes, rseed in CASES:
        if only and name not in only:
            continue
        prefix = CT.golden_prefix(name)
        r = subprocess.run(["node", os.path.join(HERE, "record_calltrace.js"), REFROOT, scene, str(w), str(h), str(k), str(passes), str(rseed), prefix],
                           capture_output=True, cwd="/tmp")
        assert r.returncode == 0, r.stderr.decode()
        sys.stderr.write(r.stderr.decode())
        t, blob = CT.load_trace(prefix)
        res = CT.play(t, blob, ref)
        acu = res["buffers"]["acu"].view(np.float32).reshape(-1, 4)
        out = {"reads": np.frombuffer(b"".join(res["reads"]), np.uint8), "read_sizes": np.array([len(x) for x in res["reads"]], np.int64),
               "sha_acu": sha(acu), "sha_seeds": sha(res["buffers"]["seeds"]), "radiance": A.radiance_sums(acu, k * k)}
        np.savez_compressed(prefix + "_expect.npz", **out)
        px = np.frombuffer(res["reads"][-1], np.uint8).reshape(-1, 4)
        print(f"{name}: {len(t['events'])} events, final frame mean rgb {px[:, :3].mean(axis=0).round(2)}, "
              f"{sum(os.path.getsize(prefix + e) for e in ('.json', '.bin.gz', '_expect.npz')) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
