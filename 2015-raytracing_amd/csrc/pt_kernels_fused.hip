// pt_kernels_fused.hip -- one launch per progressive pass.
//
// What the reference does in 44+ launches per pass (A10 code.js:1806-1854: initTrace,
// sphere/triangle/mesh closest hit, lightRender, then per segment initShadowTrace,
// any-hit kernels, sceneRender, bouncePaths ...), with every stage round-tripping
// Ray(48 B) / Poi(64 B) / shadow Ray(48 B) / acu(16 B) through memory, this kernel does
// per ray in registers: one work-item per ray, the whole path walked in one go.  HBM
// traffic is the seed (4 B in, 4 B out) and the accumulator (16 B in, 16 B out) per ray
// plus whatever geometry misses the caches; the scene description arrives as kernel
// arguments (SGPRs), geometry through the scalar/vector caches.
//
// The per-ray order of operations is exactly the order the reference's kernel sequence
// imposes on one ray id, which is what makes the result bit-identical to the granular
// path and to the oracle:
//   primary ray -> closest(spheres, triangles, mesh 0..M-1) -> lightRender(light 0..L-1)
//   -> for each light: shadow ray, any-hit(spheres, triangles, meshes), shade
//   -> `bounces` x { bounce ray, closest(...), per-light shadow + shade }
// including its quirks: lights scale `atte` once EACH (sceneRender runs per light), a
// bounce that misses re-shades the stale vertex (SURVEY 8a hazards 2, 3).
#include "pt_device.hpp"
#include "pt_launch.hpp"

namespace pt {

PT_DEV Grid mk_grid(const GridArgs& a) {
    Grid g;
    g.prims = (const float4*)a.prims;
    g.off = (const uint32_t*)a.off;
    g.bound.lo = mk3(a.bound[0], a.bound[1], a.bound[2]);
    g.bound.hi = mk3(a.bound[4], a.bound[5], a.bound[6]);
    g.n = a.n;
    return g;
}

PT_DEV void closest_all(const FusedArgs& A, Ray& ray, Poi& poi) {
    if (A.has_spheres)
        closest_hit<SPHERES>(ray, poi, mk_grid(A.spheres), nullptr, (const uint32_t*)A.spheres.matid, 0u);
    if (A.has_triangles)
        closest_hit<TRIANGLES>(ray, poi, mk_grid(A.triangles), (const float4*)A.triangles.normals,
                               (const uint32_t*)A.triangles.matid, 0u);
    for (uint32_t m = 0; m < A.n_meshes; ++m)
        closest_hit<TRIANGLES>(ray, poi, mk_grid(A.meshes[m]), (const float4*)A.meshes[m].normals, nullptr,
                               A.meshes[m].mesh_matid);
}

PT_DEV void direct_all(const FusedArgs& A, Poi& poi, int32_t& seed, float4& acc) {
    const float4* material = (const float4*)A.material;
    for (uint32_t l = 0; l < A.n_lights; ++l) {
        const LightArgs& L = A.lights[l];
        // initShadowTrace: a dead path draws nothing (code.cl:645-650)
        if (poi.matId < 0) continue;
        Ray sh = shadow_ray(poi, ld3(L.shadow), ld3(L.shadow + 3), ld3(L.shadow + 6), L.shadow[9], seed);
        if (A.has_spheres) any_hit<SPHERES>(sh, mk_grid(A.spheres));
        if (A.has_triangles) any_hit<TRIANGLES>(sh, mk_grid(A.triangles));
        for (uint32_t m = 0; m < A.n_meshes; ++m) any_hit<TRIANGLES>(sh, mk_grid(A.meshes[m]));
        if ((uint32_t)poi.matId >= A.nmat) continue;  // out-of-range id: shade nothing (see k_sceneRender)
        float4 c4 = material[poi.matId];
        f3 c = shade_vertex(poi, sh, mk3(c4.x, c4.y, c4.z), ld3(L.scene), ld3(L.scene + 3), ld3(L.scene + 6), L.scene[9]);
        acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
    }
}

__global__ void __launch_bounds__(256) k_fusedPass(const FusedArgs A) {
    const uint64_t n_local = (uint64_t)A.nrows * A.width * A.rpp;
    const uint64_t lid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lid >= n_local) return;
    const uint64_t lpix = lid / A.rpp;
    const uint32_t smp = (uint32_t)(lid - lpix * A.rpp);
    const uint32_t lrow = (uint32_t)(lpix / A.width);
    const uint32_t col = (uint32_t)(lpix - (uint64_t)lrow * A.width);
    const uint32_t row = A.row0 + lrow;

    F16 cam16;
    for (int i = 0; i < 16; ++i) cam16.v[i] = A.cam[i];
    const Cam cam = mk_cam(cam16);
    Box bound;
    bound.lo = mk3(A.bound[0], A.bound[1], A.bound[2]);
    bound.hi = mk3(A.bound[4], A.bound[5], A.bound[6]);

    int32_t seed = A.seeds[lid];
    float4 acc = ((const float4*)A.acu)[lid];

    // ---- initTrace (code.cl:458-543) for this one ray
    f3 fp = focal_point(cam, (float)col, (float)row, A.focal_length);
    float cx, cy;
    if (A.rpp > 1) {
        // un-jittered k x k lens grid; coordinates accumulate by repeated addition in the
        // reference (coord += delta), so they are rebuilt the same way
        const uint32_t side = f2u(cl_sqrt((float)A.rpp));
        const float delta = 1.0f / (float)side;
        const uint32_t i = smp / side, j = smp - i * side;
        cy = delta / 2.0f;
        for (uint32_t k = 0; k < i; ++k) cy += delta;
        cx = delta / 2.0f;
        for (uint32_t k = 0; k < j; ++k) cx += delta;
    } else {
        float2 c = ((const float2*)A.uv)[lpix];
        cx = c.x;
        cy = c.y;
    }
    Ray ray = thin_lens_ray(cam, fp, A.lens_rad, cx, cy);
    clip_to(ray, bound);
    Poi poi;
    poi.p = mk3(0.0f, 0.0f, 0.0f);
    poi.n = mk3(0.0f, 0.0f, 0.0f);
    poi.atte = mk3(1.0f, 1.0f, 1.0f);
    poi.matId = -1;

    // ---- primary segment
    closest_all(A, ray, poi);
    for (uint32_t l = 0; l < A.n_lights; ++l) {  // lightRender (code.cl:600-629)
        if (ray.mint == ray.maxt) continue;
        const LightArgs& L = A.lights[l];
        f3 irr = norm3(ld3(L.light + 6));
        if (!light_visible(ray, ld3(L.light), ld3(L.light + 3), L.light[9])) continue;
        ray.mint = PT_INF;
        ray.maxt = PT_INF;
        poi.matId = -1;
        acc.x += irr.x; acc.y += irr.y; acc.z += irr.z; acc.w += 1.0f;
    }
    direct_all(A, poi, seed, acc);

    // ---- bounces (code.js:1829-1846)
    for (uint32_t b = 0; b < A.bounces; ++b) {
        if (poi.matId >= 0) {
            ray = bounce_ray(poi, seed);
        } else {
            ray.mint = PT_INF;
            ray.maxt = PT_INF;
        }
        closest_all(A, ray, poi);
        direct_all(A, poi, seed, acc);
    }

    A.seeds[lid] = seed;
    ((float4*)A.acu)[lid] = acc;
}

void launch_fused(hipStream_t s, const FusedArgs& a) {
    const uint64_t n = (uint64_t)a.nrows * a.width * a.rpp;
    if (!n) return;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_fusedPass, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

}  // namespace pt
