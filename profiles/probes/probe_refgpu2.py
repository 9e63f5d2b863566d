"""GPU-box probe: CPU model (liboracle.so, cl_numerics.h) vs AMD's OpenCL library on the device, (1) built-in by built-in, (2) pass by pass."""
import sys, os, json, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import a10_pass as A, ref_gpu as G

def canon(a):
    u = np.ascontiguousarray(a).view(np.uint32).copy()
    u[(u & 0x7FFFFFFF) > 0x7F800000] = 0x7FC00000
    return u

orc = A.load_oracle()
orc.lib.oracle_bi_eval.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
orc.lib.oracle_bi_eval.restype = None
B = {k[2:]: v for k, v in G.BUILTIN_ARGS.items()}
mod = G.load_builtins()
rng = np.random.default_rng(7)
N = 1 << 20
spec = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 1.17549435e-38, 3.4e38, -3.4e38, 0.5, 255.0, 256.0, 2.0, 1e-20, 1e20, 1e-30, 1e30, 2147483648.0, 4294967296.0, -2147483648.0, 0.99999994, 1.0000001], np.float32)
def rnd(n, kind):
    if kind == "bits":   # any finite-ish bit pattern
        u = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32); x = u.view(np.float32).copy()
    elif kind == "unit": x = rng.uniform(-1, 1, n).astype(np.float32)
    elif kind == "ang": x = rng.uniform(-2.5, 2.5, n).astype(np.float32)
    elif kind == "wide": x = (rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
    x[:len(spec)] = spec[:min(len(spec), n)]
    return x
nvec = G.BUILTIN_SHAPES
bad_total = 0
for op, kinds in B.items():
    na, nb, nc, no = nvec.get(op, (1, 1 if len(kinds) >= 4 else 0, 1 if len(kinds) >= 5 else 0, 1))
    for dist in (["ang", "bits"] if op in ("sin", "cos") else ["unit", "wide", "bits"]):
        if op in ("sin", "cos") and dist == "bits": continue   # large arguments: Payne-Hanek path not restated
        a = rnd(N * na, dist)
        if op in ("sin", "cos"): a[np.isfinite(a) & (np.abs(a) >= 131072.0)] = 1.0   # the Payne-Hanek path is not restated
        b = rnd(N * nb, dist) if nb else None; c = rnd(N * nc, dist) if nc else None
        if b is not None: b = np.roll(b, 7)
        if c is not None: c = np.roll(c, 13)
        if op == "clamp":
            u = a.view(np.uint32); u[(u & 0x7FFFFFFF) > 0x7F800000] |= 0x00400000   # quiet NaNs only: arithmetic never makes a signalling one
        if op == "clamp": b = np.zeros(N, np.float32); c = np.full(N, rng.choice([1.0, 255.0]), np.float32)
        o_cpu = np.zeros(N * no, np.float32); o_gpu = np.zeros(N * no, np.float32)
        p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
        orc.lib.oracle_bi_eval(op.encode(), p(a), p(b), p(c), p(o_cpu), N)
        args = [mod.buf(a)] + ([mod.buf(b)] if nb else []) + ([mod.buf(c)] if nc else []) + [mod.buf(o_gpu), N]
        mod.launch("b_" + op, args, [N], [64])
        mod.flush()
        bad = np.flatnonzero(canon(o_cpu) != canon(o_gpu))
        bad_total += bad.size
        msg = ""
        if bad.size:
            i = bad[0]; j = i // no
            msg = f" first: a={a[j*na:(j+1)*na]} b={None if b is None else b[j*nb:(j+1)*nb]} c={None if c is None else c[j*nc:(j+1)*nc]} cpu={o_cpu[i]!r}({o_cpu.view(np.uint32)[i]:08x}) gpu={o_gpu[i]!r}({o_gpu.view(np.uint32)[i]:08x})"
        print(f"{op:10s} {dist:5s} mismatches {bad.size}/{N*no}{msg}", flush=True)
        mod.release()
print("BUILTINS BAD", bad_total)

k = G.GpuRefKernels()
import glob
tot_bad = 0
for f in sorted(glob.glob(os.path.join(ROOT, 'tests/golden/*.npz'))):
    name = os.path.basename(f)[:-4]
    fx = np.load(f)
    if 'seeds_in' not in fx or 'scene_json' not in fx: continue
    d = json.loads(bytes(fx["scene_json"]).decode())
    if 'rays_per_pixel' not in d or d['rays_per_pixel'] == 1: continue
    sc = A.Scene(d)
    s1 = A.PassState(sc, fx["seeds_in"]); s2 = A.PassState(sc, fx["seeds_in"])
    ck1, ck2 = {}, {}
    A.run_pass(k, sc, s1, checkpoints=ck1); A.run_pass(orc, sc, s2, checkpoints=ck2)
    k.release()
    res = {}
    for stage, (x, y) in {"primary": (ck1["primary"], ck2["primary"]), "final": (s1.snapshot(), s2.snapshot())}.items():
        live = ~(np.isinf(y["rays"]["mint"]) & np.isinf(y["rays"]["maxt"]))
        hit = y["pois"]["matId"] >= 0
        res[stage] = dict(acu=int((canon(x["acu"]) != canon(y["acu"])).sum()), seeds=int((x["seeds"] != y["seeds"]).sum()), matId=int((x["pois"]["matId"] != y["pois"]["matId"]).sum()),
                          maxt=int((canon(x["rays"]["maxt"]) != canon(y["rays"]["maxt"])).sum()), d=int((canon(x["rays"]["d"][live]) != canon(y["rays"]["d"][live])).sum()),
                          p=int((canon(x["pois"]["p"][hit]) != canon(y["pois"]["p"][hit])).sum()), n=int((canon(x["pois"]["normal"][hit]) != canon(y["pois"]["normal"][hit])).sum()),
                          atte=int((canon(x["pois"]["atte"]) != canon(y["pois"]["atte"])).sum()), sh=int((canon(x["shadow"]["maxt"]) != canon(y["shadow"]["maxt"])).sum()))
    res["pixel"] = int((s1.pixel != s2.pixel).sum())
    b = sum(sum(v.values()) if isinstance(v, dict) else v for v in res.values())
    tot_bad += b
    print(name, sc.total_rays, "rays:", "IDENTICAL" if b == 0 else res, flush=True)
print("PASS BAD", tot_bad)
