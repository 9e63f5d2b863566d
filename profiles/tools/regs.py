#!/usr/bin/env python3
"""Resource usage (VGPRs, scratch, occupancy, spills, static LDS) of the kernels of one csrc/*.hip file, from the compiler's own
-Rpass-analysis=kernel-resource-usage remarks.  usage: regs.py [file.hip] [-Dflags...]   (ALL=1: every kernel, not only k_fusedPass)"""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "2015-raytracing_amd", "csrc")
args = sys.argv[1:]
src = args.pop(0) if args and args[0].endswith(".hip") else "pt_kernels_fused.hip"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/regs_tmp.o"] + args
out = subprocess.run(cmd, cwd=here, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
pats = {"sgpr": r"TotalSGPRs: (\d+)", "vgpr": r" VGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)",
        "sspill": r"SGPRs Spill: (\d+)", "vspill": r"VGPRs Spill: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)"}
cur = None
for l in out.splitlines():
    if "error" in l:
        print(l)
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        continue
    if cur is None:
        continue
    for k, p in pats.items():
        m = re.search(p, l)
        if m:
            cur[k] = m.group(1)
    if "lds" in cur:
        if "k_fusedPass" in cur["name"] or os.environ.get("ALL") == "1":
            n = cur["name"].replace("_ZN2pt11k_fusedPassILb", "<").replace("EEEvNS_9FusedArgsEPjPKjj", ">").replace("ELi", ",")
            print("%-30s vgpr %3s sgpr %3s scratch %3s occ %s sspill %2s vspill %2s lds %s" % (n, cur["vgpr"], cur["sgpr"], cur["scratch"], cur["occ"], cur["sspill"], cur["vspill"], cur["lds"]))
        cur = None
