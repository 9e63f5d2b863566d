#!/usr/bin/env python3
"""profiles/tools/isa_blocks.py [kernel-substring] -- basic blocks of one fused-pass kernel (default k_fusedPass<true,0,0>) with their VALU / SALU /
LDS / SMEM / VMEM instruction counts and the loop nest the compiler's comments give them: the static half of "where do the instructions go"
(profiles/trip_counts.py is the dynamic half).  Compiles pt_kernels_fused.hip to assembly with the product flags (+ any -D given)."""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "2015-raytracing_amd", "csrc")
args = sys.argv[1:]
want = args.pop(0) if args and not args[0].startswith("-") else "k_fusedPassILb1ELi0ELi0E"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-S", "--cuda-device-only", "pt_kernels_fused.hip", "-o", "/tmp/isa_blocks.s"] + args
subprocess.run(cmd, cwd=here, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
lines = open("/tmp/isa_blocks.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2pt11" + want + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (a kernel has several s_endpgm: its early returns)
body = lines[start:end + 1]
blocks, cur = [], {"label": "entry", "depth": 0, "ins": [], "line": 0}
for n, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "depth": 0, "ins": [], "line": n}
        continue
    m = re.search(r"Depth[= ](\d+)", l)
    if m and ("Loop Header" in l or "in Loop" in l):
        cur["depth"] = max(cur["depth"], int(m.group(1)))
    m = re.match(r"^; %bb\.(\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": "bb." + m.group(1), "depth": 0, "ins": [], "line": n}
        m2 = re.search(r"Depth=(\d+)", l)
        if m2: cur["depth"] = int(m2.group(1))
        continue
    t = l.strip()
    if t and not t.startswith(";") and not t.startswith("."):
        cur["ins"].append(t.split()[0])
blocks.append(cur)
def cls(op):
    if op.startswith(("v_cmp", "v_cndmask", "v_mov", "v_readfirstlane", "v_readlane", "v_writelane", "v_accvgpr")): return "vmisc"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("s_"): return "salu"
    return "other"
tot = {}
print(f"{'line':>5s} {'block':12s} depth  valu vmisc salu lds smem vmem")
for b in blocks:
    c = {}
    for op in b["ins"]:
        c[cls(op)] = c.get(cls(op), 0) + 1
        tot[cls(op)] = tot.get(cls(op), 0) + 1
    if not b["ins"]: continue
    print(f"{b['line']:5d} {b['label']:12s} {b['depth']:5d} {c.get('valu',0):5d} {c.get('vmisc',0):5d} {c.get('salu',0):4d} {c.get('lds',0):3d} {c.get('smem',0):4d} {c.get('vmem',0):4d}")
print("total", tot, "lines", len(body))
