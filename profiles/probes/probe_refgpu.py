import sys, os, json, numpy as np
sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import a10_pass as A, ref_gpu as G
k = G.GpuRefKernels()
print("sizeofRay", k.sizeofRay(), "sizeofPoi", k.sizeofPoi())
for name in ["cornell_32x24_r4","cornell_teapot3_32x24_r4","basic_32x24_r4","cornell_16x12_r9"]:
    fx = np.load(f"tests/golden/{name}.npz")
    sc = A.Scene(json.loads(bytes(fx["scene_json"]).decode()))
    seeds = fx["seeds_in"]
    st = A.PassState(sc, seeds)
    A.run_pass(k, sc, st)
    same_mat = (st.pois["matId"] == fx["f_pois_matId"]).mean()
    d = st.acu - fx["f_acu"]
    print(name, "matId agree", same_mat, "acu rms", float(np.sqrt((d.astype(np.float64)**2).mean())), "max", float(np.abs(d).max()), "seeds equal", np.array_equal(st.seeds, fx["f_seeds"]), "pixel eq", np.array_equal(st.pixel, fx["pixel"]), (st.pixel.astype(int)-fx["pixel"]).__abs__().max())
    orc = A.load_oracle(); st2 = A.PassState(sc, seeds); A.run_pass(orc, sc, st2)
    print("   oracle(cpu, old contract) vs fixture acu equal:", np.array_equal(st2.acu, fx["f_acu"]))
    k.release()
