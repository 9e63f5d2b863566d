"""PackedScene: the kernel inputs of one scene, as numpy arrays.

The dict it wraps is what the JavaScript host emits with `node host/cli.js pack <scene.xml>` (and what
tests/golden/*.npz carry as `scene_json`): camera / light float16 packs, material float4s, grid-sorted
sphere / triangle / mesh arrays with their cell offsets and AABBs -- the buffers the reference host
uploads in prepare*() (A10 code.js:1156-1291, 1364-1379).
"""
import json

import numpy as np


class PackedScene:
    def __init__(self, d):
        if isinstance(d, (str, bytes)):
            d = json.loads(d)
        self.d = d
        self.width, self.height, self.rpp = int(d["width"]), int(d["height"]), int(d["rays_per_pixel"])
        self.n_slabs = int(d.get("n_slabs", 1))
        f32 = lambda k: np.asarray(d[k], dtype=np.float32)
        u32 = lambda k: np.asarray(d[k], dtype=np.uint32)
        self.cam, self.bounds = f32("cam"), f32("bounds")
        self.focal_length, self.lens_rad = float(d["focal_length"]), float(d["lens_rad"])
        self.has_spheres = int(d.get("n_spheres", 0)) > 0
        self.has_triangles = int(d.get("n_triangles", 0)) > 0
        if self.has_spheres:
            self.spheres, self.s_matid, self.s_box, self.sphere_bounds = f32("spheres"), u32("s_matid"), u32("s_box"), f32("sphere_bounds")
        if self.has_triangles:
            self.t_pos, self.t_normal, self.t_matid, self.t_box = f32("t_pos"), f32("t_normal"), u32("t_matid"), u32("t_box")
            self.triangle_bounds = f32("triangle_bounds")
        self.meshes = [dict(pos=np.asarray(m["pos"], np.float32), normal=np.asarray(m["normal"], np.float32),
                            box=np.asarray(m["box"], np.uint32), matid=int(m["matid"]),
                            bounds=np.asarray(m["bounds"], np.float32), nslabs=int(m["nslabs"])) for m in d.get("meshes", [])]
        self.lights = [dict(shadow=np.asarray(l["shadow"], np.float32), scene=np.asarray(l["scene"], np.float32),
                            light=np.asarray(l["light"], np.float32)) for l in d["lights"]]
        self.materials = f32("materials")

    @property
    def total_rays(self):
        return self.width * self.height * self.rpp

    def resized(self, width, height, rpp):
        """Same scene at another image size / sample count: only the camera pack depends on them
        (Camera.lookAt, A10 code.js:203-217: width = height * cols/rows; cols, rows in .sE/.sF)."""
        d = dict(self.d)
        cam = list(np.asarray(d["cam"], np.float64))
        cam[12] = float(np.float32(cam[13] * (width / height)))
        cam[14], cam[15] = float(width), float(height)
        d.update(cam=cam, width=width, height=height, rays_per_pixel=rpp)
        return PackedScene(d)
