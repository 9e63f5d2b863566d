#!/usr/bin/env python3
"""bench.py -- Msamples/s of the Assign10 path-tracing pass on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], the one the metric is quoted on): scenes/cornell.xml packed by the
reference host for 1920x1080, rays_per_pixel = 256 (16x16 lens grid), one progressive pass, depth 8 (eight
bounces after the primary hit; the reference hard-codes five, A10 code.js:1829 -- `--bounces 5` measures that,
and the default run reports it beside the headline in "depth5"), seeds s[id] = 1 + (mix32(id ^ 0x9E3779B9)
mod 2147483646) generated on the device.  One "step" is one such frame: initTrace .. 8 bounces .. copyToPixel,
through mirt_render_first_pass: ONE fused launch that also resolves the pixels (round 4: no per-ray accumulator, 8 B per sample + 20 B per
pixel of HBM traffic; --keep-acu: the accumulator written per ray as a progressive second pass would need it).  Beside the headline the
default run reports the grid kernel (`grid_scene`: cornell_teapot3) and BASELINE configs 2 / 3 (`frames`), each with a roofline fraction and a
bounded CPU baseline, and the same frame on the second numerics contract (`default_contract`: libmirt_default.so, in a child process) (--no-extras skips them).
Inputs (scene buffers, seeds) are resident in HBM before the timed region.

N > 1: one process per GPU; the frame is cut into N contiguous row tiles (ray ids stay global, so the
image is identical for every N); each step ends with the one real exchange of the path, an RCCL
all_gather of the RGBA8 tiles.  Total work is fixed -> "scaling": "strong".

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (pt::k_fusedPass): the binding
roof of this path is the fp32 vector ALU, so `bound` is "valu" with the 157.3 TFLOP/s
vector peak; the HBM view the north star asks for rides along in `roofline_hbm`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

# measured constants (DESIGN.md "Algorithmic work"): `make -C oracle count && python oracle/count_flops.py` -- the CPU oracle with its
# operation counters on the scene at 320x240 x 16 rays (+ - * / sqrt sin cos = 1, a fused multiply-add = 2); {scene: {bounces: flop/sample}}
FLOPS_PER_SAMPLE = {"cornell": {5: 6612.3, 8: 9788.3}, "cornell_teapot3": {5: 10905.6, 8: 16342.6}}
# --scene NAME: tests/golden/<this fixture>.npz carries the scene as the reference host packed it
SCENE_FIXTURE = {"cornell_teapot3": "cornell_teapot3_32x24_r4", "cornell_official": "cornell_official_64x48_r1", "basic": "basic_32x24_r4",
                 "triangles": "triangles_32x24_r4", "twoLights": "twoLights_32x24_r4", "threeLights": "threeLights_32x24_r1",
                 "own_gems": "own_gems_48x36_r4", "own_studio": "own_studio_48x36_r4", "own_flat": "own_flat_32x24_r4"}
# Configs 2 and 3 (SURVEY 8d "flops = W*H*sum(tests by exit)"): algorithmic fp32 operations per pixel of the reference's own nested loops, measured at the
# configs' full sizes by the counting oracle (make -C oracle count && python oracle/count_flops_frames.py)
FRAME_FLOPS_PER_PIXEL = {"a04_parliament_1024": 260428.3, "a04_teapot_1024": 28400.0,
                         "a07_parliament_1080p_n2": 8085.5, "a07_parliament_1080p_n16": 211.4, "a07_parliament_1080p_n32": 118.9}
BYTES_PER_SAMPLE_SEED = 8.0         # seed 4 in + 4 out: all a sample moves when the pass resolves its pixels itself
BYTES_PER_SAMPLE_ACU = 16.0         # + the per-ray accumulator written (--keep-acu, or a ray count that does not divide 256)
BYTES_PER_PIXEL_RESOLVE = 4.0 + 16.0   # RGBA8 + the fp32 radiance sums
PEAK_VALU_TFLOPS = 157.3            # MI355X_MICROARCH.md: peak FP32 vector (FMA-counted)
PEAK_HBM_GBS = 8000.0
SIMDS, NOMINAL_HZ, MEASURED_ISSUE_CYCLES = 1024, 2.4e9, 2.4   # profiles/micro/valu_rate.hip: cycles per plain fp32 VOP2 wave-instruction at 8 waves (spec: 2)


def resolves_in_pass(rpp):
    """ray counts at which a frame's first pass resolves its own pixels (include/mirt.h mirt_render_first_pass): divisors of 256, or 256 times a power of two up to 32"""
    if rpp <= 256:
        return 256 % rpp == 0
    c = rpp // 256
    return rpp % 256 == 0 and c <= 32 and c & (c - 1) == 0


def csrc_sha256():
    """Hash of the kernel sources this libmirt.so is built from: the stamp that ties a rocprofv3 PMC summary to a build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "2015-raytracing_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.cpp")) + glob.glob(os.path.join(d, "*.sh"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()


def find_pmc_summary(path, workload_key, kernel_name):
    """Counters for the bench line come from a FILE a rocprofv3 run wrote (profiles/run_profile.sh + summarize.py), never from constants:
    `path` if given, else the newest profiles/*/pmc_summary.json whose stamp names THIS source tree and THIS workload.  Returns
    (path, per-kernel counters) or (None, reason)."""
    import glob
    want = csrc_sha256()
    cands = [path] if path else sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.json")), key=os.path.getmtime, reverse=True)
    why = "no profiles/*/pmc_summary.json carries a stamp for this source tree and workload"
    for c in cands:
        try:
            d = json.load(open(c))
        except (OSError, ValueError) as e:
            why = f"{c}: {e}"
            continue
        st = d.get("_stamp") or {}
        if st.get("csrc_sha256") != want or st.get("workload") != workload_key:
            if path:
                why = f"{c}: stamped for csrc {str(st.get('csrc_sha256'))[:12]} / {st.get('workload')}, this run is {want[:12]} / {workload_key}"
            continue
        squash = lambda t: t.replace(" ", "")      # rocprofv3 writes "void pt::k_fusedPass<true, 0>"
        k = next((v for name, v in d.items() if squash(kernel_name) in squash(name)), None)
        if k is None:
            why = f"{c}: no counters for {kernel_name}"
            continue
        return os.path.relpath(c, ROOT), k
    return None, why


def cpu_baseline(packed_json, log, bounces, rpp=256):
    """The CPU oracle (our plain-C restatement, OpenMP) on a bounded sample.  Threads = the CPUs this process may really use: the
    affinity mask capped by the cgroup CPU quota (oracle/a10_pass.py cpu_budget); `cores` reports that number."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import a10_pass as A
    d = json.loads(packed_json)
    w, h = 480, 270
    cam = list(d["cam"])
    cam[14], cam[15] = float(w), float(h)   # same 16:9 frustum, 1/4 of the pixels per side
    d.update(cam=cam, width=w, height=h, rays_per_pixel=rpp)
    sc = A.Scene(d)
    k = A.load_oracle()
    k.lib.oracle_num_threads.restype = int
    cores = k.lib.oracle_num_threads()
    st = A.PassState(sc, A.make_seeds(sc.total_rays))
    t0 = time.perf_counter()
    A.run_pass(k, sc, st, bounces=bounces)
    dt = time.perf_counter() - t0
    log(f"cpu_baseline: {sc.total_rays} samples in {dt:.2f} s on {cores} threads")
    return {"value": round(sc.total_rays / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "host_cpus_visible": os.cpu_count(), "kind": "port",
            "sample": f"the same scene at {w}x{h} rpp{rpp} 1 pass {bounces} bounces ({sc.total_rays} samples, {dt:.1f} s wall)",
            "note": "the CPU checker models v_rsq_f32 / v_sqrt_f32 through 2 x 16 MB measured tables (oracle/cl_numerics.h): bit-exact, cache-unfriendly"}


def reference_gpu_baseline(packed_json, log, bounces):
    """The same baseline leg, second half: the REFERENCE'S OWN kernels -- its code.cl as AMD's OpenCL toolchain compiles it for gfx950
    (oracle/_ref/a10_gfx950.hsaco, a checker; present where the build container's tree travelled) -- driven through executeRender's
    launch sequence on this very GPU (oracle/ref_gpu.py), on a bounded sample: the frame at 16 rays per pixel.  What the reference itself
    achieves on an MI355X, beside `value`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import a10_pass as A
    import ref_gpu as G
    if not G.available():
        return None
    d = json.loads(packed_json)
    d.update(rays_per_pixel=16)
    sc = A.Scene(d)
    n, npix = sc.total_rays, sc.width * sc.height
    k = G.GpuRefKernels()

    class St:
        pass
    st = St()
    st.rays, st.pois, st.shadow = G.DevBuf(n * 48), G.DevBuf(n * 64), G.DevBuf(n * 48)
    st.acu, st.seeds, st.pixel = G.DevBuf(n * 16), G.DevBuf(n * 4), G.DevBuf(npix * 4)
    st.passes = 1
    st.seeds.upload(A.make_seeds(n))
    A.run_pass(k, sc, st, bounces=bounces)
    G.chk(G.hip().hipDeviceSynchronize(), "sync")
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        A.run_pass(k, sc, st, bounces=bounces, init_acu=False)
    G.chk(G.hip().hipDeviceSynchronize(), "sync")
    dt = (time.perf_counter() - t0) / reps
    for b in (st.rays, st.pois, st.shadow, st.acu, st.seeds, st.pixel):
        b.free()
    k.release()
    log(f"reference_gpu_baseline: {n} samples in {dt * 1e3:.1f} ms per pass")
    return {"value": round(n / dt / 1e6, 1), "unit": "Msamples/s", "kind": "reference", "where": "this MI355X",
            "what": "the reference's code.cl compiled by AMD's OpenCL toolchain for gfx950, its 14 kernels launched in executeRender's order",
            "sample": f"the same scene at {sc.width}x{sc.height} rpp16 1 pass {bounces} bounces ({n} samples, {dt * 1e3:.1f} ms per pass)"}


def grid_scene_record(ctx, log, render, scene, cpu):
    """The grid kernel beside the headline: cornell_teapot3.xml (992-triangle teapot in 10^3 cells + a 20-triangle box in 5^3, two lights: the scene class
    seven of the reference's ten A10 scenes belong to) at 1920x1080 x 16 rays per pixel, five bounces (the reference's depth) and eight."""
    name = "cornell_teapot3"
    fxs = np.load(os.path.join(ROOT, "tests", "golden", SCENE_FIXTURE[name] + ".npz"))
    sc = scene.PackedScene(bytes(fxs["scene_json"]).decode()).resized(1920, 1080, 16)
    fr = render.FusedRenderer(ctx, sc, want_radiance=True, keep_acu=False)
    out = {"workload": f"A10 {name}.xml path trace 1920x1080, 16 spp, 1 pass; fused mirt_render_first_pass (k_fusedPass<true,1,*>: shared-test grid walk)",
           "samples": fr.nrays}
    for bounces in (5, 8):
        ms = []
        for i in range(4):
            ctx.seed_fill(fr.seeds, fr.first_ray, fr.nrays, 0)
            fr.passes = 1
            fr.execute_render(bounces=bounces, fresh=True)
            if i:
                ms.append(ctx.pass_timing()[0])
        t = float(np.mean(ms))
        fl = FLOPS_PER_SAMPLE[name][bounces]
        out[f"depth{bounces}"] = {"launch_ms": round(t, 3), "Msamples_per_s_kernel": round(fr.nrays / t / 1e3, 1), "flops_per_sample": fl,
                                 "achieved_TFLOPs": round(fl * fr.nrays / (t * 1e-3) / 1e12, 2), "frac": round(fl * fr.nrays / (t * 1e-3) / 1e12 / PEAK_VALU_TFLOPS, 4)}
    fr.release()
    if cpu:
        c = cpu_baseline(json.dumps(sc.d), log, 5, 16)
        out["cpu_baseline"] = c
    return out


def frames_record(ctx, log, render, cpu):
    """BASELINE configs 2 and 3 at full size: Assign04 brute force (house_of_parliament, 9 144 triangles, and teapot, 992) at 1024 x 1024; Assign07
    uniform-grid traversal of house_of_parliament at 1920 x 1080, n_slabs 2 (the page's default) / 16 / 32 (2 and 32 binned on the device, mirt_grid_build).
    Kernel time = HIP events around the trace kernel alone; flops = the reference's own nested loops' operations (FRAME_FLOPS_PER_PIXEL)."""
    def job(name):
        fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        return json.loads(bytes(fx["frame_json"]).decode())
    a04, a07, tea = job("frame_a04_parliament_96x64"), job("frame_a07_parliament_n16_160x120"), job("frame_a04_teapot_160x120")
    jobs = [("a04_parliament_1024", render.frame_resized(a04, 1024, 1024)), ("a04_teapot_1024", render.frame_resized(tea, 1024, 1024))]
    for n in (2, 16, 32):
        jobs.append((f"a07_parliament_1080p_n{n}", render.frame_resized(a07 if n == 16 else render.frame_regrid(ctx, a07, a04, n), 1920, 1080)))
    out = {}
    for tag, d in jobs:
        fp = render.FramePacked(d)
        render.render_frame(ctx, fp)   # warm-up: prepares the triangles, validates the cell table
        ms = []
        for _ in range(5):
            t = {}
            render.render_frame(ctx, fp, timing=t)
            ms.append(t["trace_ms"])
        k_ms = float(np.median(ms))
        npx = d["width"] * d["height"]
        fl = FRAME_FLOPS_PER_PIXEL[tag] * npx
        rec = {"kernel": "pt::k_a04_meshTrace" if d["assign"] == 4 else "pt::k_a07_meshTrace", "kernel_ms": round(k_ms, 4), "kernel_ms_min": round(min(ms), 4),
               "Mrays_per_s": round(npx / k_ms / 1e3, 1), "flops_per_pixel": FRAME_FLOPS_PER_PIXEL[tag],
               "achieved_TFLOPs": round(fl / (k_ms * 1e-3) / 1e12, 2), "frac": round(fl / (k_ms * 1e-3) / 1e12 / PEAK_VALU_TFLOPS, 4)}
        if d["assign"] == 4:
            rec["pairs_per_s_T"] = round(npx * d["t_size"] / (k_ms * 1e-3) / 1e12, 2)
        out[tag] = rec
    out["note"] = ("flops are the REFERENCE's work (its nested loops, every test to its own exit); the kernels skip whole groups of 16 triangles whose bounding sphere "
                   "every ray of the wave misses, bit-identical pixels -- so `frac` prices the reference's operations against the kernel's time and may exceed 1 "
                   "(a04_parliament, a07 n = 2): it is a speed-up over doing the reference's arithmetic at peak, not a utilisation")
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import frame_pass as F
        base = {}
        for tag, d, (w, h) in (("a04_parliament", a04, (256, 256)), ("a07_parliament_n16", a07, (960, 540))):
            fr = F.Frame(render.frame_resized(d, w, h))
            t0 = time.perf_counter()
            F.run_frame("oracle", fr)
            dt = time.perf_counter() - t0
            base[tag] = {"value": round(w * h / dt / 1e6, 3), "unit": "Mrays/s", "kind": "port", "sample": f"{w}x{h}, {dt:.2f} s wall"}
        import a10_pass as A
        k = A.load_oracle()
        k.lib.oracle_num_threads.restype = int
        out["cpu_baseline"] = dict(base, cores=k.lib.oracle_num_threads())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--rpp", type=int, default=None, help="rays per pixel (a square); default 256 for the headline scene, 16 with --scene")
    ap.add_argument("--scene", default="cornell", help="cornell (BASELINE configs[3], the default) or another A10 scene the fixtures carry: "
                    "cornell_teapot3 (two grid meshes, two lights), own_gems, twoLights, ...")
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--no-depth5", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the sub-records beside the headline: grid_scene (cornell_teapot3) and frames (configs 2 / 3)")
    ap.add_argument("--keep-acu", action="store_true", help="keep the 16-byte per-ray accumulator (what a progressive second pass needs); default: the one-pass "
                    "frame without it -- the pass resolves its pixels itself, 8 B per sample + 20 B per pixel (SURVEY 8d)")
    ap.add_argument("--pmc-summary", default=None, help="pmc_summary.json of a rocprofv3 run of this command (profiles/run_profile.sh); default: the "
                    "newest one under profiles/ stamped with this source tree and workload.  Without one, `traffic` and `issue` are null")
    args = ap.parse_args()
    if args.rpp is None:
        args.rpp = 256 if args.scene == "cornell" else 16

    # stdout carries exactly one line, the JSON: native libraries that print banners to fd 1 (RCCL does at init) go to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the RCCL path is exercised even with one rank, so `--nproc-per-node 1` rehearses
    # exactly the code the N-GPU runs execute; a bare `python bench.py` stays free of any process group
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def log(msg):
        if rank == 0:
            print(msg, file=sys.stderr, flush=True)

    graft.load_package()
    from raytracing_amd.pyhost import mirt, render, scene, tiling

    name = f"scene_cornell_{args.width}x{args.height}_r{args.rpp}.json"
    path = os.path.join(ROOT, "tests", "golden", name)
    if args.scene != "cornell":
        if args.scene not in SCENE_FIXTURE:
            raise SystemExit(f"--scene {args.scene}: known scenes are cornell, " + ", ".join(sorted(SCENE_FIXTURE)))
        fxs = np.load(os.path.join(ROOT, "tests", "golden", SCENE_FIXTURE[args.scene] + ".npz"))
        packed = json.dumps(scene.PackedScene(bytes(fxs["scene_json"]).decode()).resized(args.width, args.height, args.rpp).d)
        sc = scene.PackedScene(packed)
    elif os.path.exists(path):
        packed = open(path).read()
        sc = scene.PackedScene(packed)
    else:  # other sizes: same scene, camera re-packed for the size
        packed = open(os.path.join(ROOT, "tests", "golden", "scene_cornell_1920x1080_r256.json")).read()
        sc = scene.PackedScene(packed).resized(args.width, args.height, args.rpp)

    # contiguous row tiles
    row0, nrows = tiling.row_tiles(sc.height, world)[rank]
    max_rows = tiling.padded_rows(sc.height, world)

    ctx = mirt.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_profiling(True)
    if os.environ.get("MIRT_EXACT_ONLY") == "1":   # A/B knob: the single exact kernel instead of the default optimistic pair
        ctx.set_exact_only(True)
    no_acu = not (args.keep_acu or os.environ.get("BENCH_KEEP_ACU") == "1") and resolves_in_pass(sc.rpp) and os.environ.get("MIRT_INPASS_RESOLVE", "1") != "0"
    fr = render.FusedRenderer(ctx, sc, row0=row0, nrows=nrows, want_radiance=True, keep_acu=not no_acu)
    # the RGBA8 tile lives in a torch tensor so RCCL can move it
    tile = torch.zeros(max_rows * sc.width * 4, dtype=torch.uint8, device="cuda")
    fr.pixel.release()
    fr.pixel = ctx.wrap(tile.data_ptr(), max_rows * sc.width * 4)
    frame = torch.empty(world * tile.numel(), dtype=torch.uint8, device="cuda") if use_dist else None

    gather_ev = []   # (start, stop) torch events around each step's all_gather, on the stream the kernels run on

    def step():
        # a step re-renders the same frame: restore the seeds it started from; the accumulator is initialised by the pass itself
        # (mirt_render_first_pass = preRender's initAcu folded into the first pass, A10 code.js:1078-1099)
        ctx.seed_fill(fr.seeds, fr.first_ray, fr.nrays, 0)
        fr.passes = 1
        fr.execute_render(bounces=args.bounces, fresh=True)
        if use_dist:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tiling.gather_tiles(tile, frame)
            e1.record()
            gather_ev.append((e0, e1))

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    gather_ev.clear()
    kern_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kern_ms.append(ctx.pass_timing())   # waits for this step's own events only
    fence()
    dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())

    # N > 1: every rank's own kernel time and its gather's own event time (the all_gather includes waiting for the slowest rank), so that a
    # first run on N real GPUs is diagnosable from its one line: load imbalance shows in launch_ms, transport trouble in gather_ms
    per_rank = None
    if use_dist:
        mine = torch.tensor([float(np.mean([k[0] for k in kern_ms])), float(np.mean([a.elapsed_time(b) for a, b in gather_ev])) if gather_ev else 0.0,
                             float(nrows)], dtype=torch.float64, device="cuda")
        allr = torch.empty(world * 3, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(allr, mine)
        a = allr.view(world, 3).cpu().numpy()
        per_rank = {"launch_ms": [round(float(x), 3) for x in a[:, 0]], "gather_ms": [round(float(x), 3) for x in a[:, 1]], "rows": [int(x) for x in a[:, 2]],
                    "gather": "torch.distributed all_gather_into_tensor (RCCL) of padded RGBA8 tiles, %d bytes per rank" % tile.numel()}
    samples_per_step = sc.width * sc.height * sc.rpp
    value = samples_per_step * args.steps / dt / 1e6
    fused_ms = float(np.mean([k[0] for k in kern_ms]))
    resolve_ms = float(np.mean([k[1] for k in kern_ms]))
    local_samples = fr.nrays
    fps_table = FLOPS_PER_SAMPLE.get(args.scene)
    flops = None if fps_table is None else fps_table.get(args.bounces, fps_table[5] * (1 + args.bounces) / 6.0)
    valu_tf = (flops or 0.0) * local_samples / (fused_ms * 1e-3) / 1e12
    has_grids = any(m["nslabs"] > 1 for m in sc.d.get("meshes", [])) or sc.d.get("n_slabs", 1) > 1
    # <optimistic, grids: 0 none / 1 tables in LDS / 2 in memory, waves: 0 the default build / 5 the grid kernels' 5-wave build>; matched as a prefix
    kernel_name = "pt::k_fusedPass<true,1," if has_grids else "pt::k_fusedPass<true,0,"
    bytes_per_sample = BYTES_PER_SAMPLE_SEED + (BYTES_PER_PIXEL_RESOLVE / sc.rpp if no_acu else BYTES_PER_SAMPLE_ACU)
    hbm_gbs = bytes_per_sample * local_samples / (fused_ms * 1e-3) / 1e9

    # counters: only what a named rocprofv3 summary of this build and workload holds (FETCH_SIZE / WRITE_SIZE in KiB, separate passes;
    # gfx950 halves FETCH_SIZE on wide coalesced reads: MI355X_MICROARCH.md)
    workload_key = f"{args.scene}_{sc.width}x{sc.height}_r{sc.rpp}_b{args.bounces}_n{world}"
    pmc_path, pmc = find_pmc_summary(args.pmc_summary, workload_key, kernel_name) if rank == 0 else (None, "rank != 0")
    kernel_name += "*>" if has_grids else "0>"   # the name as rocprofv3 prints it, spaces aside (the grid kernels come in two occupancy builds)
    cnt = (lambda c: pmc[c]["mean_per_dispatch"] if c in pmc else None) if pmc_path else (lambda c: None)
    fetch, write = cnt("FETCH_SIZE"), cnt("WRITE_SIZE")
    traffic = (fetch * 2 + write) * 1024.0 if fetch is not None and write is not None else None
    valu, waves = cnt("SQ_INSTS_VALU"), cnt("SQ_WAVES")
    valu_per_sample = valu / waves if valu and waves else None   # wave-instructions per wave = per sample-lane
    out = {
        "metric": "Msamples/sec (pixels x spp) at 1920x1080", "value": round(value, 2), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "fps": round(args.steps / dt, 3),
        "config": {"workload": f"A10 {args.scene}.xml path trace {sc.width}x{sc.height}, {sc.rpp} spp ({int(round(sc.rpp ** 0.5))}x{int(round(sc.rpp ** 0.5))} lens grid), 1 pass, "
                               f"{args.bounces} bounces, thin lens; " + ("one fused launch that resolves its own pixels (mirt_render_first_pass, no per-ray accumulator)" if no_acu else "fused mirt_render_first_pass + copyToPixel")
                               + (f"; {world} row tiles + RCCL all_gather of RGBA8" if world > 1 else ""),
                   "width": sc.width, "height": sc.height, "rays_per_pixel": sc.rpp, "bounces": args.bounces,
                   "parallelism": f"rows/{world}", "workload_key": workload_key},
        "roofline": {"kernel": kernel_name, "bound": "valu", "achieved": round(valu_tf, 2) if flops else None, "peak": PEAK_VALU_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(valu_tf / PEAK_VALU_TFLOPS, 4) if flops else None, "traffic": traffic,
                     "traffic_source": pmc_path if traffic is not None else None,
                     "flops_per_sample": flops, "launch_ms": round(fused_ms, 3), "resolve_ms": round(resolve_ms, 3)},
        "issue": None if valu_per_sample is None else {
            "what": "VALU wave-instructions issued per second by the kernel (SQ_INSTS_VALU / SQ_WAVES of the named summary x this run's samples / its launch time) "
                    "vs what 1024 SIMDs sustain on plain fp32 VOP2 streams",
            "valu_instr_per_sample": round(valu_per_sample, 1),
            "achieved_Ginstr_s": round(valu_per_sample * local_samples / 64.0 / (fused_ms * 1e-3) / 1e9, 1),
            "attainable_Ginstr_s": round(SIMDS * NOMINAL_HZ / MEASURED_ISSUE_CYCLES / 1e9, 1), "spec_Ginstr_s": round(SIMDS * NOMINAL_HZ / 2.0 / 1e9, 1),
            "frac_of_attainable": round(valu_per_sample * local_samples / 64.0 / (fused_ms * 1e-3) / (SIMDS * NOMINAL_HZ / MEASURED_ISSUE_CYCLES), 4),
            "source": pmc_path + ", profiles/micro/README.md"},
        "roofline_hbm": {"kernel": kernel_name, "bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": round(hbm_gbs / PEAK_HBM_GBS, 5), "traffic": traffic,
                         "traffic_note": (f"algorithmic {bytes_per_sample * local_samples / 1e9:.2f} GB/launch (seeds in + out, " + ("RGBA8 + radiance per pixel" if no_acu else "accumulator out") + "); measured "
                                          f"{fetch * 2 * 1024 / 1e9:.2f} + {write * 1024 / 1e9:.2f} GB (FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes, {pmc_path})") if traffic is not None
                                         else f"no counters: {pmc if not pmc_path else 'summary lacks FETCH_SIZE / WRITE_SIZE'}",
                         "bytes_per_sample": round(bytes_per_sample, 3), "resolve": "in the pass (no per-ray accumulator)" if no_acu else "separate copyToPixel over acu"},
        "build": {"csrc_sha256": csrc_sha256()[:16]},
    }
    if per_rank:
        out["per_rank"] = per_rank
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(packed, log, args.bounces, sc.rpp)
        try:
            rg = reference_gpu_baseline(packed, log, args.bounces)
        except Exception as e:   # a baseline, never a reason to lose the bench line
            log(f"reference_gpu_baseline skipped: {e}")
            rg = None
        if rg:
            out["reference_gpu_baseline"] = rg
    if rank == 0 and world == 1 and args.bounces == 8 and not args.no_depth5 and fps_table:
        # the reference's own depth (five bounces), same frame, two untimed-warmup-free steps: reported beside the headline
        t5 = []
        for _ in range(2):
            ctx.seed_fill(fr.seeds, fr.first_ray, fr.nrays, 0)
            fr.passes = 1
            fr.execute_render(bounces=5, fresh=True)
            t5.append(ctx.pass_timing()[0])
        out["depth5"] = {"launch_ms": round(float(np.mean(t5)), 3), "Msamples_per_s_kernel": round(fr.nrays / np.mean(t5) / 1e3, 1),
                         "flops_per_sample": fps_table[5], "frac": round(fps_table[5] * fr.nrays / (np.mean(t5) * 1e-3) / 1e12 / PEAK_VALU_TFLOPS, 4)}
    fr.release()
    if rank == 0 and world == 1 and args.scene == "cornell" and not args.no_extras:
        # beside the headline: the grid kernel and BASELINE configs 2 / 3, each with its own roofline fraction and a bounded CPU baseline
        try:
            out["grid_scene"] = grid_scene_record(ctx, log, render, scene, not args.no_cpu)
            out["frames"] = frames_record(ctx, log, render, not args.no_cpu)
        except Exception as e:   # sub-records never cost the headline line
            log(f"extras skipped: {type(e).__name__}: {e}")
            out["extras_error"] = f"{type(e).__name__}: {e}"
        if os.environ.get("MIRT_CONTRACT") != "default" and os.path.exists(os.path.join(ROOT, "2015-raytracing_amd", "libmirt_default.so")):
            # the SECOND numerics contract -- libmirt_default.so, the kernels built as the reference's own host builds its program (2.5-ulp division,
            # DESIGN.md section 2) -- on the same frame: a process of its own (a process loads one libmirt), started as a child, three timed steps
            try:
                import subprocess
                env = dict(os.environ, MIRT_CONTRACT="default")
                env.pop("MIRT_LIB_PATH", None)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu", "--no-extras", "--no-depth5"],
                                   env=env, capture_output=True, text=True, timeout=240)
                d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["default_contract"] = {"library": "libmirt_default.so", "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                                           "launch_ms": d["roofline"]["launch_ms"], "frac": d["roofline"]["frac"],
                                           "note": "the reference as its own host builds it (program.build() without options); bit-exact against that build: tests/test_default_contract.py"}
            except Exception as e:
                log(f"default-contract record skipped: {type(e).__name__}: {e}")
    ctx.destroy()
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
