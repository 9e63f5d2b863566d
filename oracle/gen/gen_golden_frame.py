#!/usr/bin/env python3
"""oracle/gen/gen_golden_frame.py -- TEST INFRASTRUCTURE, build container only.

tests/golden/frame_*.npz for BASELINE configs 1-3 (Assign01 / Assign04 / Assign07), from the REFERENCE run here:
inputs = the reference's own A01/A04/A07 host code in a Node vm sandbox (ref_host_dump_frame.js), outputs = its own
code.cl compiled for x86 (oracle/Makefile `ref`).  Run: make -C oracle ref && python oracle/gen/gen_golden_frame.py
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frame_pass as F  # noqa: E402

REFROOT = os.environ.get("REFROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
OWN_PAGE = os.path.join(ROOT, "tests", "scenes", "page")

# (name, assign, mesh, width, height, n_slabs, own page?)
CASES = [
    ("frame_a01_64x48", 1, "-", 64, 48, 0, False),
    ("frame_a01_512x512", 1, "-", 512, 512, 0, False),                       # BASELINE config 1
    ("frame_a04_teapot_160x120", 4, "teapot.json", 160, 120, 0, False),
    ("frame_a04_parliament_96x64", 4, "house_of_parliament.json", 96, 64, 0, False),   # config 2's mesh, small frame
    ("frame_a07_teapot_n2_160x120", 7, "teapot.json", 160, 120, 2, False),   # n_slabs 2 = the page's default
    ("frame_a07_teapot_n8_160x120", 7, "teapot.json", 160, 120, 8, False),
    ("frame_a07_parliament_n16_160x120", 7, "house_of_parliament.json", 160, 120, 16, False),   # config 3's mesh
    ("frame_a04_house_160x120", 4, "house.json", 160, 120, 0, False),        # the third selectable mesh of the A04 / A07 pages
    ("frame_a07_house_n2_160x120", 7, "house.json", 160, 120, 2, False),
    ("frame_a07_house_n8_160x120", 7, "house.json", 160, 120, 8, False),
    ("frame_a07_mol_benzene_n2_96x64", 7, "benzene.pdb", 96, 64, 2, False),      # molecule mode, the page's default n_slabs
    ("frame_a07_mol_c60_n4_160x120", 7, "c60.pdb", 160, 120, 4, False),
    ("frame_a07_mol_dna_n8_160x120", 7, "dna.pdb", 160, 120, 8, False),
    ("frame_a07_mol_3IZ4_n16_96x64", 7, "3IZ4.pdb", 96, 64, 16, False),          # the page's largest molecule: the sphere walk at scale
    ("frame_a07_own_mol_helix_n3_96x64", 7, "helix.pdb", 96, 64, 3, True),       # every reader quirk (tests/scenes/make_molecules.py)
    ("frame_a07_own_mol_lattice_n6_96x64", 7, "lattice.pdb", 96, 64, 6, True),
    ("frame_a04_own_icosphere_96x64", 4, "icosphere.json", 96, 64, 0, True),
    ("frame_a07_own_terrain_n5_96x64", 7, "terrain.json", 96, 64, 5, True),
    ("frame_a07_own_octahedra_n3_96x64", 7, "octahedra.json", 96, 64, 3, True),
]


def main():
    only = set(sys.argv[1:])
    for name, assign, mesh, w, h, n, own in CASES:
        if only and name not in only:
            continue
        cmd = ["node", os.path.join(HERE, "ref_host_dump_frame.js"), REFROOT, str(assign), mesh, str(w), str(h), str(n)]
        if own:
            cmd.append(OWN_PAGE)
        js = subprocess.run(cmd, check=True, capture_output=True, cwd="/tmp").stdout.decode()
        fr = F.Frame(json.loads(js))
        pixels, rays = F.run_frame("ref", fr)
        out = {"frame_json": np.frombuffer(js.encode(), np.uint8), "pixel": pixels}
        if rays is not None:
            out["rays_maxt"] = np.ascontiguousarray(rays["maxt"])
            out["rays_mint"] = np.ascontiguousarray(rays["mint"])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: lit {float((pixels[:, :3].max(axis=1) > 0).mean()):.3f}, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
