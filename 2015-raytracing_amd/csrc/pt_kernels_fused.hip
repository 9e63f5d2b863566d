// pt_kernels_fused.hip -- one launch per progressive pass.
//
// What the reference does in 44+ launches per pass (A10 code.js:1806-1854: initTrace,
// sphere/triangle/mesh closest hit, lightRender, then per segment initShadowTrace,
// any-hit kernels, sceneRender, bouncePaths ...), with every stage round-tripping
// Ray(48 B) / Poi(64 B) / shadow Ray(48 B) / acu(16 B) through memory, this kernel does
// per ray in registers: one work-item per ray, the whole path walked in one go.  HBM
// traffic is the seed (4 B in, 4 B out) and the accumulator (16 B in, 16 B out) per ray;
// the scene description arrives as kernel arguments (SGPRs) and the geometry, for the
// single-cell grids of loose primitives (n_slabs == 1, A10 code.js:399), through
// wave-uniform loops, i.e. scalar loads shared by the 64 lanes of a wave.
//
// The per-ray order of operations is exactly the order the reference's kernel sequence
// imposes on one ray id, which is what makes the result bit-identical to the granular
// path and to the oracle:
//   primary ray -> closest(spheres, triangles, mesh 0..M-1) -> lightRender(light 0..L-1)
//   -> for each light: shadow ray, any-hit(spheres, triangles, meshes), shade
//   -> `bounces` x { bounce ray, closest(...), per-light shadow + shade }
// including its quirks: lights scale `atte` once EACH (sceneRender runs per light), a
// bounce that misses re-shades the stale vertex (SURVEY 8a hazards 2, 3).
//
// Triangles are read from a PREPARED copy of the host's position buffer (k_prepTriangles):
// {p0, e1 = p1-p0, e2 = p2-p0, n = cross(e2,e1)} -- the ray-independent head of
// Moeller-Trumbore (A10 code.cl:252-256), computed once with the same fp32 operations, so
// every value that reaches a ray-dependent operation has the bits it has in the reference.
#include "pt_device.hpp"
#include "pt_launch.hpp"

#ifndef PT_OPT_PREFETCH
#define PT_OPT_PREFETCH 0
#endif

namespace pt {

// prepared triangle: 3 x float4 = {p0.xyz, n.x} {e1.xyz, n.y} {e2.xyz, n.z}
__global__ void __launch_bounds__(256) k_prepTriangles(const float4* pos, float4* out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    f3 p0 = ld3(pos[3u * i]), p1 = ld3(pos[3u * i + 1]), p2 = ld3(pos[3u * i + 2]);
    f3 e1 = sub3(p1, p0);
    f3 e2 = sub3(p2, p0);
    f3 n = cross3(e2, e1);
    out[3u * i] = make_float4(p0.x, p0.y, p0.z, n.x);
    out[3u * i + 1] = make_float4(e1.x, e1.y, e1.z, n.y);
    out[3u * i + 2] = make_float4(e2.x, e2.y, e2.z, n.z);
}

struct Hit { uint32_t idx; float t, beta, gamma; };

// Moeller-Trumbore on a prepared triangle; same operations, same order and the same
// accept/reject predicates as inter_triangle (pt_device.hpp), written without early exits:
// in a wave-uniform loop the 64 lanes leave at different tests anyway.
PT_DEV bool tri_test(f3 o, f3 d, float cmin, float cmax, const float4 A, const float4 B, const float4 C,
                     float& t_out, float& beta_out, float& gamma_out) {
    const f3 p0 = mk3(A.x, A.y, A.z), e1 = mk3(B.x, B.y, B.z), e2 = mk3(C.x, C.y, C.z), n = mk3(A.w, B.w, C.w);
    float div = dot3(n, d);
    float idiv = 1.0f / div;
    f3 s = sub3(o, p0);
    float beta = dot3(cross3(s, d), e2) * idiv;
    float gamma = dot3(cross3(s, e1), d) * idiv;
    float gb = gamma + beta;
    float t = dot3(cross3(s, e2), e1) * -idiv;
    bool ok = !(div <= 0);
    ok = ok && !(beta < 0.0f || beta > 1.0f);
    ok = ok && !(gamma < 0.0f || gb < 0.0f || gb > 1.0f);
    ok = ok && (t >= cmin && t <= cmax);
    t_out = t;
    beta_out = beta;
    gamma_out = gamma;
    return ok;
}

struct SphereRay { float a, inv2a; };  // ray-only part of the quadratic (A10 code.cl:203, 218)
PT_DEV SphereRay sphere_ray(f3 d) {
    SphereRay r;
    r.a = dot3(d, d);
    r.inv2a = 1.0f / (2.0f * r.a);
    return r;
}
PT_DEV bool sph_test(f3 o, f3 d, const SphereRay& sr, float cmin, float cmax, const float4 sph, float& t_out) {
    f3 omc = sub3(o, ld3(sph));
    float b = 2.0f * dot3(omc, d);
    float c = dot3(omc, omc) - sph.w;
    float dis = cl_mad(-4.0f * c, sr.a, b * b);
    float sq = cl_sqrt(dis);
    float t0 = (-b - sq) * sr.inv2a;
    float t1 = (-b + sq) * sr.inv2a;
    float tmin = cl_fmin(t0, t1);
    float tmax = cl_fmax(t0, t1);
    const bool in0 = (tmin >= cmin && tmin <= cmax);
    const bool in1 = (tmax >= cmin && tmax <= cmax);
    t_out = in0 ? tmin : tmax;
    return !(dis < 0.0f) && (in0 || in1);
}

// One primitive set.  KIND / ANY as in pt_device.hpp.  n == 1: a single cell, every lane walks
// the same list -> wave-uniform loop, scalar loads.  n > 1: per-lane 3-axis DDA.
template <int KIND, bool ANY>
PT_DEV Hit trace_set(const Ray& ray, const BoxHit& bh, const GridArgs& S) {
    const float4* __restrict__ prims = (const float4*)S.prims;
    const uint32_t* __restrict__ off = (const uint32_t*)S.off;
    Hit ch;
    ch.idx = UINT32_MAX;
    ch.t = ray.maxt;
    ch.beta = 0.0f;
    ch.gamma = 0.0f;
    SphereRay sr;
    if (KIND == SPHERES) sr = sphere_ray(ray.d);

    if (S.n == 1u) {
        // axis_setup with n == 1: slab = 0, the cell exit is the far face as the reference computes
        // it, lo + (0 + (d>=0)) * ((hi-lo)/1)   (A10 code.cl:699-707)
        float tn[3];
        if (S.exit_is_far_face) {  // host-verified: the cell's exit planes ARE the box's far planes (see mirt_abi.cpp)
            tn[0] = bh.tfx; tn[1] = bh.tfy; tn[2] = bh.tfz;
        } else {
            const float lo[3] = {S.bound[0], S.bound[1], S.bound[2]}, hi[3] = {S.bound[4], S.bound[5], S.bound[6]};
            const float oo[3] = {ray.o.x, ray.o.y, ray.o.z}, dd[3] = {ray.d.x, ray.d.y, ray.d.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float delta = (hi[k] - lo[k]) / 1.0f;
                float xnext = lo[k] + (float)((dd[k] >= 0) ? 1 : 0) * delta;
                tn[k] = (xnext - oo[k]) / dd[k];
            }
        }
        const float cmin = bh.tmin;
        const float cmax = cl_min(cl_min(tn[0], tn[1]), tn[2]);
        const uint32_t begin = __builtin_amdgcn_readfirstlane(off[0]);
        const uint32_t end = __builtin_amdgcn_readfirstlane(off[1]);
        bool done = false;
#if PT_OPT_PREFETCH
        // software pipeline: the next primitive's scalar loads are issued before the current test's ~60 VALU ops
        float4 A = make_float4(0, 0, 0, 0), B = A, C = A;
        if (begin < end) {
            if (KIND == SPHERES) A = prims[begin];
            else { A = prims[3u * begin]; B = prims[3u * begin + 1]; C = prims[3u * begin + 2]; }
        }
        for (uint32_t i = begin; i < end; ++i) {
            float4 nA = A, nB = B, nC = C;
            if (i + 1 < end) {
                if (KIND == SPHERES) nA = prims[i + 1];
                else { nA = prims[3u * (i + 1)]; nB = prims[3u * (i + 1) + 1]; nC = prims[3u * (i + 1) + 2]; }
            }
            float ti, b = 0.0f, gm = 0.0f;
            bool hit;
            if (KIND == SPHERES) hit = sph_test(ray.o, ray.d, sr, cmin, cmax, A, ti);
            else hit = tri_test(ray.o, ray.d, cmin, cmax, A, B, C, ti, b, gm);
            const bool better = !done && hit && ti < ch.t;
            if (better) { ch.t = ti; ch.idx = i; ch.beta = b; ch.gamma = gm; }
            if (ANY) {
                done = done || better;
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            }
            A = nA; B = nB; C = nC;
        }
#else
        for (uint32_t i = begin; i < end; ++i) {
            float ti, b = 0.0f, gm = 0.0f;
            bool hit;
            if (KIND == SPHERES) {
                hit = sph_test(ray.o, ray.d, sr, cmin, cmax, prims[i], ti);
            } else {
                hit = tri_test(ray.o, ray.d, cmin, cmax, prims[3u * i], prims[3u * i + 1], prims[3u * i + 2], ti, b, gm);
            }
            const bool better = !done && hit && ti < ch.t;
            if (better) { ch.t = ti; ch.idx = i; ch.beta = b; ch.gamma = gm; }
            if (ANY) {
                done = done || better;
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;  // every lane of the wave is blocked
            }
        }
#endif
        return ch;
    }

    Axis ax = axis_setup(ray.o.x, ray.d.x, bh.tmin, S.bound[0], S.bound[4], S.n);
    Axis ay = axis_setup(ray.o.y, ray.d.y, bh.tmin, S.bound[1], S.bound[5], S.n);
    Axis az = axis_setup(ray.o.z, ray.d.z, bh.tmin, S.bound[2], S.bound[6], S.n);
    float t = bh.tmin;
    const uint32_t zs = S.n * S.n, ys = S.n;
    for (;;) {
        const float cmin = t;
        const float cmax = cl_min(cl_min(ax.tnext, ay.tnext), az.tnext);
        const uint32_t cell = (uint32_t)az.slab * zs + (uint32_t)ay.slab * ys + (uint32_t)ax.slab;
        const uint32_t begin = off[cell], end = off[cell + 1];
        for (uint32_t i = begin; i < end; ++i) {
            float ti, b = 0.0f, gm = 0.0f;
            bool hit;
            if (KIND == SPHERES) {
                hit = sph_test(ray.o, ray.d, sr, cmin, cmax, prims[i], ti);
            } else {
                hit = tri_test(ray.o, ray.d, cmin, cmax, prims[3u * i], prims[3u * i + 1], prims[3u * i + 2], ti, b, gm);
            }
            if (hit && ti < ch.t) {
                ch.t = ti; ch.idx = i; ch.beta = b; ch.gamma = gm;
                if (ANY) break;
            }
        }
        if (ch.idx != UINT32_MAX) break;
        t = cmax;
        if (t == ax.tnext) {
            ax.tnext += ax.dt;
            if (t >= bh.tmax) break;
            ax.slab += ax.dslab;
            if (ax.slab == ax.limit) break;
        } else if (t == ay.tnext) {
            ay.tnext += ay.dt;
            if (t >= bh.tmax) break;
            ay.slab += ay.dslab;
            if (ay.slab == ay.limit) break;
        } else {
            az.tnext += az.dt;
            if (t >= bh.tmax) break;
            az.slab += az.dslab;
            if (az.slab == az.limit) break;
        }
    }
    return ch;
}

PT_DEV Box set_box(const GridArgs& S) {
    Box b;
    b.lo = mk3(S.bound[0], S.bound[1], S.bound[2]);
    b.hi = mk3(S.bound[4], S.bound[5], S.bound[6]);
    return b;
}

// closest hit over every set in upload order, z-buffered through ray.maxt
// (A10 code.cl:675-800, 802-935, 937-1070; order A10 code.js:1809-1813)
PT_DEV void closest_all(const FusedArgs& A, Ray& ray, Poi& poi) {
    for (uint32_t s = 0; s < A.n_sets; ++s) {
        const GridArgs& S = A.sets[s];
        if (ray.mint == ray.maxt) continue;
        BoxHit bh = inter_aabb(ray, set_box(S));
        if (!bh.v) continue;
        if (S.kind == KIND_SPHERES) {
            Hit ch = trace_set<SPHERES, false>(ray, bh, S);
            if (ch.idx == UINT32_MAX) continue;
            ray.maxt = ch.t;
            poi.p = add3(ray.o, scl3(ch.t, ray.d));
            poi.n = norm3(sub3(poi.p, ld3(((const float4*)S.prims)[ch.idx])));
            poi.matId = (int32_t)((const uint32_t*)S.matid)[ch.idx];
        } else {
            Hit ch = trace_set<TRIANGLES, false>(ray, bh, S);
            if (ch.idx == UINT32_MAX) continue;
            ray.maxt = ch.t;
            poi.p = add3(ray.o, scl3(ch.t, ray.d));
            const float4* nn = (const float4*)S.normals + 3u * (size_t)ch.idx;
            float w = 1.0f - ch.beta - ch.gamma;  // code.cl:409-411
            poi.n = norm3(add3(add3(scl3(w, ld3(nn[0])), scl3(ch.beta, ld3(nn[1]))), scl3(ch.gamma, ld3(nn[2]))));
            poi.matId = (int32_t)(S.matid ? ((const uint32_t*)S.matid)[ch.idx] : S.mesh_matid);
        }
    }
}

// per light: shadow ray, any-hit over every set, shade (A10 code.js:1817-1826; code.cl:631-673,
// 1073-1321, 1323-1364)
PT_DEV void direct_all(const FusedArgs& A, Poi& poi, int32_t& seed, float4& acc) {
    const float4* material = (const float4*)A.material;
    for (uint32_t l = 0; l < A.n_lights; ++l) {
        const LightArgs& L = A.lights[l];
        if (poi.matId < 0) continue;  // initShadowTrace: a dead path draws nothing (code.cl:645-650)
        Ray sh = shadow_ray(poi, ld3(L.shadow), ld3(L.shadow + 3), ld3(L.shadow + 6), L.shadow[9], seed);
        for (uint32_t s = 0; s < A.n_sets; ++s) {
            const GridArgs& S = A.sets[s];
            if (sh.mint == sh.maxt) continue;
            BoxHit bh = inter_aabb(sh, set_box(S));
            if (!bh.v) continue;
            Hit ch = (S.kind == KIND_SPHERES) ? trace_set<SPHERES, true>(sh, bh, S) : trace_set<TRIANGLES, true>(sh, bh, S);
            sh.maxt = ch.t;
            if (ch.idx != UINT32_MAX) sh.mint = ch.t;
        }
        if ((uint32_t)poi.matId >= A.nmat) continue;  // out-of-range id: shade nothing (see k_sceneRender)
        float4 c4 = material[poi.matId];
        f3 c = shade_vertex(poi, sh, mk3(c4.x, c4.y, c4.z), ld3(L.scene), ld3(L.scene + 3), ld3(L.scene + 6), L.scene[9]);
        acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
    }
}

#ifndef PT_OPT_PREFETCH
#define PT_OPT_PREFETCH 0
#endif
#ifndef PT_FUSED_WAVES
#define PT_FUSED_WAVES 6   // waves per SIMD the register allocator must leave room for (A/B: 4 -> 213 ms, 5 -> 205, 6 -> 200, 8 -> 243 with spills)
#endif
__global__ void __launch_bounds__(256, PT_FUSED_WAVES) k_fusedPass(const FusedArgs A) {
    const uint64_t n_local = (uint64_t)A.nrows * A.width * A.rpp;
    const uint64_t lid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lid >= n_local) return;
    const uint64_t lpix = lid / A.rpp;
    const uint32_t smp = (uint32_t)(lid - lpix * A.rpp);
    const uint32_t lrow = (uint32_t)(lpix / A.width);
    const uint32_t col = (uint32_t)(lpix - (uint64_t)lrow * A.width);
    const uint32_t row = A.row0 + lrow;

    Cam cam;
    cam.eye = ld3(A.cam); cam.U = ld3(A.cam + 3); cam.V = ld3(A.cam + 6); cam.W = ld3(A.cam + 9);
    cam.width = A.cam[12]; cam.height = A.cam[13];
    cam.cols = f2u(A.cam[14]); cam.rows = f2u(A.cam[15]);
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

    // segment 0 is the primary ray; segments 1..bounces start with bouncePaths (code.js:1829-1846)
    for (uint32_t seg = 0; seg <= A.bounces; ++seg) {
        if (seg > 0) {
            if (poi.matId >= 0) {
                ray = bounce_ray(poi, seed);
            } else {
                ray.mint = PT_INF;
                ray.maxt = PT_INF;
            }
        }
        closest_all(A, ray, poi);
        if (seg == 0) {
            for (uint32_t l = 0; l < A.n_lights; ++l) {  // lightRender (code.cl:600-629), primary segment only
                if (ray.mint == ray.maxt) continue;
                const LightArgs& L = A.lights[l];
                f3 irr = norm3(ld3(L.light + 6));
                if (!light_visible(ray, ld3(L.light), ld3(L.light + 3), L.light[9])) continue;
                ray.mint = PT_INF;
                ray.maxt = PT_INF;
                poi.matId = -1;
                acc.x += irr.x; acc.y += irr.y; acc.z += irr.z; acc.w += 1.0f;
            }
        }
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

void launch_prepTriangles(hipStream_t s, const void* pos, void* out, uint32_t count) {
    if (!count) return;
    hipLaunchKernelGGL(k_prepTriangles, dim3((count + 255) / 256), dim3(256), 0, s, (const float4*)pos, (float4*)out, count);
}

}  // namespace pt
