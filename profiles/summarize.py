#!/usr/bin/env python3
"""profiles/summarize.py <gpurun_out/prof_TAG> <profiles/DIR> [workload_key] -- condense one run_profile.sh output into
kernel_stats.csv + pmc_summary.json (mean per dispatch, per kernel, per counter) and print the fused kernel's line.
The summary is STAMPED with the hash of the kernel sources it was measured on and the workload (run_profile.sh records both on
the GPU box, in stamp.json): bench.py only quotes counters from a summary whose stamp matches the build it is running."""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(os.path.join(src, "trace/*/*_kernel_stats.csv"))[0], os.path.join(dst, "kernel_stats.csv"))
out = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mix", "pmc_sq3", "pmc_sqc"):
    fs = glob.glob(os.path.join(src, d, "*/*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.setdefault(r["Kernel_Name"].split("(")[0], {})["VGPR_Count"] = int(r["VGPR_Count"])
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: {"mean_per_dispatch": sum(x) / len(x), "dispatches": len(x)} for c, x in v.items()})
stamp = os.path.join(src, "stamp.json")
if os.path.exists(stamp):
    out["_stamp"] = json.load(open(stamp))
    if len(sys.argv) > 3:
        out["_stamp"]["workload"] = sys.argv[3]
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
f = next((v for k, v in out.items() if "k_fusedPass<true" in k), out.get("pt::k_fusedPass", {}))
if "_stamp" in out:
    print("stamp:", out["_stamp"])
g = lambda c: f.get(c, {}).get("mean_per_dispatch", float("nan"))
waves = g("SQ_WAVES")
print(open(os.path.join(dst, "kernel_stats.csv")).read().split("\n")[1])
print(f"k_fusedPass: VALU/wave {g('SQ_INSTS_VALU')/waves:.0f}  SALU/wave {g('SQ_INSTS_SALU')/waves:.0f}  VMEM_RD/wave {g('SQ_INSTS_VMEM_RD')/waves:.1f}  "
      f"SMEM/wave {g('SQ_INSTS_SMEM')/waves:.1f}  lane-util {g('SQ_THREAD_CYCLES_VALU')/(g('SQ_INSTS_VALU')*64):.3f}  "
      f"FETCH {g('FETCH_SIZE')*2*1024/1e9:.2f} GB(x2)  WRITE {g('WRITE_SIZE')*1024/1e9:.2f} GB  VGPR {f.get('VGPR_Count')}")
