#!/usr/bin/env python3
"""tests/scenes/make_meshes.py -- writes the small triangle meshes of our own test page
(tests/scenes/page/tri/*.json) in the Assimp-style JSON layout the reference's mesh parser reads:
meshes[].{vertexPositions, vertexNormals, indices, materialIndex}, nodes[].{modelMatrix, meshIndices},
materials[].diffuseReflectance.  Geometry is ours (an icosphere, an octahedron pair, a terrain patch)."""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def icosphere(subdiv):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    nrm = lambda p: tuple(c / math.sqrt(sum(q * q for q in p)) for c in p)
    v = [nrm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                v.append(nrm(tuple((v[a][i] + v[b][i]) / 2 for i in range(3))))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return v, f


def mesh(v, f, normals, mat):
    # the reference's triangle test is single-sided: it accepts dot(cross(e2,e1), d) > 0, i.e. faces whose
    # (p0,p1,p2) winding is counter-clockwise seen from the ray's side -- the usual outward-CCW convention
    return {"vertexPositions": [round(c, 6) for p in v for c in p], "vertexNormals": [round(c, 6) for p in normals for c in p],
            "indices": [i for a, b, c in f for i in (a, b, c)], "materialIndex": mat}


def rot_y(deg, s=1.0, t=(0, 0, 0)):
    c, si = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    # column-major 4x4 (gl-matrix layout)
    return [round(x, 6) for x in (c * s, 0, -si * s, 0, 0, s, 0, 0, si * s, 0, c * s, 0, t[0], t[1], t[2], 1)]


def main():
    out = os.path.join(HERE, "page", "tri")
    os.makedirs(out, exist_ok=True)
    v, f = icosphere(1)  # 80 triangles
    ico = {"name": "icosphere", "materials": [{"diffuseReflectance": [0.8, 0.7, 0.2, 1]}],
           "meshes": [mesh(v, f, v, 0)], "nodes": [{"modelMatrix": rot_y(25, 1.3, (0.1, -0.2, 0.05)), "meshIndices": [0]}]}
    json.dump(ico, open(os.path.join(out, "icosphere.json"), "w"))
    ov = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    of = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    octa = {"name": "octahedra", "materials": [{"diffuseReflectance": [0.2, 0.4, 0.9, 1]}, {"diffuseReflectance": [0.9, 0.3, 0.3, 1]}],
            "meshes": [mesh(ov, of, ov, 0), mesh(ov, of, ov, 1)],
            "nodes": [{"modelMatrix": rot_y(0, 0.5, (-0.8, 0, 0)), "meshIndices": [0]}, {"modelMatrix": rot_y(40, 0.7, (0.6, 0.2, -0.3)), "meshIndices": [1]}]}
    json.dump(octa, open(os.path.join(out, "octahedra.json"), "w"))
    # un-indexed terrain patch without nodes (exercises the identity-matrix branch): 6x6 quads of a height field z = h(x, y) that faces
    # the frame camera (which sits on +z and looks down -z, A07 code.js:55-71), with the field's own normals so the shade varies
    P, N, n = [], [], 6
    h = lambda x, y: 0.15 * math.sin(3 * x) * math.cos(2 * y)
    def nrm(x, y):
        gx, gy = 0.45 * math.cos(3 * x) * math.cos(2 * y), -0.30 * math.sin(3 * x) * math.sin(2 * y)
        l = math.sqrt(gx * gx + gy * gy + 1.0)
        return (-gx / l, -gy / l, 1.0 / l)
    for i in range(n):
        for j in range(n):
            x0, x1, y0, y1 = i / n - 0.5, (i + 1) / n - 0.5, j / n - 0.5, (j + 1) / n - 0.5
            q = [(x0, y0, h(x0, y0)), (x1, y0, h(x1, y0)), (x1, y1, h(x1, y1)), (x0, y1, h(x0, y1))]
            for tri in ((0, 1, 2), (0, 2, 3)):   # counter-clockwise seen from +z -> visible to the camera
                for k in tri:
                    P += [round(c, 6) for c in q[k]]
                    N += [round(c, 6) for c in nrm(q[k][0], q[k][1])]
    terr = {"name": "terrain", "materials": [{"diffuseReflectance": [0.4, 0.8, 0.4, 1]}],
            "meshes": [{"vertexPositions": P, "vertexNormals": N, "indices": list(range(len(P) // 3)), "materialIndex": 0}]}
    json.dump(terr, open(os.path.join(out, "terrain.json"), "w"))
    print("wrote", os.listdir(out))


if __name__ == "__main__":
    main()
