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
#include <stdlib.h>
#include "pt_trace_coop.hpp"

#ifndef PT_SKIP_DARK_SHADOWS
#define PT_SKIP_DARK_SHADOWS 1   // grid kernels: a shadow ray whose vertex the light cannot light is not traced (direct_all)
#endif
#ifndef PT_AABB_UNSIGNED_ZERO
#define PT_AABB_UNSIGNED_ZERO 1   // single-cell sets only (their tmin / tmax / exits are compare-only): see slab1_fast
#endif

#ifndef PT_LANE_LISTS
#define PT_LANE_LISTS 1          // optimistic kernel: single-cell triangle sets through per-lane candidate lists (pt_trace.hpp trace_cell1, LANES)
#endif
#ifndef PT_LANE_LISTS_GRIDS
#define PT_LANE_LISTS_GRIDS 1    // ... in the grid kernels as well (cornell_teapot3 35.8 -> 33.4 ms, cornell_teapot 24.0 -> 22.7, own_gems 12.4 -> 12.8)
#endif
#define PT_LANE_LISTS_FOR(FAST, GRIDS) ((FAST) && PT_LANE_LISTS && ((GRIDS) == 0 || PT_LANE_LISTS_GRIDS))

namespace pt {

// prepared triangle: 3 x float4 = {p0.xyz, n.x} {e1.xyz, n.y} {e2.xyz, n.z}
// `insane` (one word, zeroed by the caller) is set when a plane-normal component is neither zero nor within [2^-40, 2^40]:
// such a set must not use the fast reciprocal forms (GridArgs::fast_ok).
__global__ void __launch_bounds__(256) k_prepTriangles(const float4* pos, float4* out, uint32_t count, uint32_t* insane) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    f3 p0 = ld3(pos[3u * i]), p1 = ld3(pos[3u * i + 1]), p2 = ld3(pos[3u * i + 2]);
    f3 e1 = sub3(p1, p0);
    f3 e2 = sub3(p2, p0);
    f3 n = cross3(e2, e1);
    out[3u * i] = make_float4(p0.x, p0.y, p0.z, n.x);
    out[3u * i + 1] = make_float4(e1.x, e1.y, e1.z, n.y);
    out[3u * i + 2] = make_float4(e2.x, e2.y, e2.z, n.z);
    auto bad = [](float v) { const float a = __builtin_fabsf(v); return !(v == 0.0f || (a >= 9.094947e-13f && a <= 1.0995116e12f)); };
    // ... and every vertex / edge component within 2^21 in magnitude (NaN fails): the bounds of the set are the caller's word, the
    // finiteness arguments of the optimistic kernel (pt_trace.hpp) are about the triangles themselves
    auto big = [](float v) { return !(__builtin_fabsf(v) <= 2097152.0f); };
    if (bad(n.x) || bad(n.y) || bad(n.z) || big(p0.x) || big(p0.y) || big(p0.z) || big(e1.x) || big(e1.y) || big(e1.z) || big(e2.x) || big(e2.y) || big(e2.z))
        atomicOr(insane, 1u);
    // One bounding sphere per group of kTriGroup consecutive records, behind the records: {centre, R'^2} with R' the radius inflated by
    // 1 % plus what the centre's own rounding can move it (pt_trace.hpp group_missed).  The frame kernels of Assign04 / 07 skip a group
    // whose sphere the ray's line misses.
    if ((i % kTriGroup) == 0u) {
        const uint32_t last = i + kTriGroup < count ? i + kTriGroup : count;
        f3 lo = p0, hi = p0;
        for (uint32_t k = 3u * i; k < 3u * last; ++k) {
            const f3 v = ld3(pos[k]);
            lo = mk3(__builtin_fminf(lo.x, v.x), __builtin_fminf(lo.y, v.y), __builtin_fminf(lo.z, v.z));
            hi = mk3(__builtin_fmaxf(hi.x, v.x), __builtin_fmaxf(hi.y, v.y), __builtin_fmaxf(hi.z, v.z));
        }
        const f3 c = mk3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
        float r2 = 0.0f;
        for (uint32_t k = 3u * i; k < 3u * last; ++k) {
            const f3 v = sub3(ld3(pos[k]), c);
            r2 = __builtin_fmaxf(r2, v.x * v.x + v.y * v.y + v.z * v.z);
        }
        const float cm = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c.x), __builtin_fabsf(c.y)), __builtin_fabsf(c.z));
        const float r = __builtin_sqrtf(r2) * 1.01f + 1e-5f * cm + 1e-30f;
        // (a NaN coordinate is ignored by min / max: the triangle it belongs to can only produce a NaN t, which the reference's own
        // interval test rejects -- skipping it changes nothing; an infinite one makes the radius infinite: never skipped)
        ((float4*)(out + 3u * (size_t)count))[i / kTriGroup] = make_float4(c.x, c.y, c.z, r * r);
    }
}

// The candidate sweep's plane list of a small set (count <= kLdsTriMax; GridArgs::pnorm), from the prepared records (layout: pt_launch.hpp
// "plane list").  Per chunk of 32 records (one candidate word), one entry per PLANE -- every record of the chunk whose plane is the same bit
// for bit (n = cross(e2, e1) with -0 read as +0, and k = p0 . n; for an axis plane the coordinate and the one non-zero normal component)
// shares it through a 32-bit record mask (record c0 + j at bit 31 - j): the halves of a quad wherever they sit in the list.  A record whose
// plane is the same with the normal REVERSED (-n, -k: the back face of a double-sided triangle) rides in the entry's second mask: the sweep
// gets its verdict from two more fused operations instead of a whole evaluation.  AXIS planes -- n has two zero components and both edges
// are exactly zero along the third axis, so the plane is {x_a = p0_a} exactly -- are listed per axis as {q = p0_a, n_a, mask, back mask}: the
// sweep needs one product for n . d and two operations for n . o - k.  General planes: {n.xyz, k, mask, back mask, 0, 0}.  Entries travel
// in 64-byte groups (one s_load_dwordx16: four axis entries or two general ones), each class padded with zero-mask entries.
// Margin constants (pt_trace.hpp): M = G |o|_1 + H, G = 2^-17 |e1|_1 |e2|_1, H = max(G |p0|_1, 2^-56), every product rounded up; the sweep
// uses the chunk's largest G and H for every plane of the chunk (a wider margin keeps more, never less).  Serial: at most 96 records, once per buffer content.
__global__ void __launch_bounds__(64) k_planeList(const float4* prep, uint32_t count, uint32_t* out) {
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    uint32_t* hdr = out;
    uint32_t* ent = out + 16;   // the header is 64 bytes; groups of 16 words follow
    const float up = 1.0000002384185791015625f;   // 1 + 2^-22: more than the rounding of the sums and products below
    auto canon = [](float v) { return v == 0.0f ? 0u : __float_as_uint(v); };   // -0 -> +0 (the sign of a zero component cannot change what n . d <= 0 decides)
    uint32_t groups = 0;   // 64-byte groups written so far
    for (uint32_t c = 0; c < 4u; ++c) {
        const uint32_t lo = c * 32u, hi = lo + 32u < count ? lo + 32u : count;
        hdr[c] = 0u; hdr[4u + c] = groups; hdr[8u + c] = 0u; hdr[12u + c] = 0u;
        if (lo >= count) continue;
        float gmax = 0.0f, hmax = 0.0f;
        // class of every record of the chunk: 0 / 1 / 2 = axis plane along x / y / z, 3 = general; and its key
        uint32_t cls[32], key[32][4];
        for (uint32_t i = lo; i < hi; ++i) {
            const float4 A = prep[3u * i], B = prep[3u * i + 1], C = prep[3u * i + 2];
            const float k = (float)((double)A.x * A.w + (double)A.y * B.w + (double)A.z * C.w);
            const float E = ((__builtin_fabsf(B.x) + __builtin_fabsf(B.y) + __builtin_fabsf(B.z)) * up) * ((__builtin_fabsf(C.x) + __builtin_fabsf(C.y) + __builtin_fabsf(C.z)) * up) * up;
            const float G = 0x1p-17f * E;
            const float H = __builtin_fmaxf(G * ((__builtin_fabsf(A.x) + __builtin_fabsf(A.y) + __builtin_fabsf(A.z)) * up) * up, 0x1p-56f);
            gmax = __builtin_fmaxf(gmax, G); hmax = __builtin_fmaxf(hmax, H);
            const float n[3] = {A.w, B.w, C.w}, p0[3] = {A.x, A.y, A.z}, e1[3] = {B.x, B.y, B.z}, e2[3] = {C.x, C.y, C.z};
            uint32_t cl = 3u;
            for (uint32_t a = 0; a < 3u; ++a)
                if (n[a] != 0.0f && n[(a + 1u) % 3u] == 0.0f && n[(a + 2u) % 3u] == 0.0f && e1[a] == 0.0f && e2[a] == 0.0f) cl = a;
            cls[i - lo] = cl;
            if (cl < 3u) { key[i - lo][0] = canon(p0[cl]); key[i - lo][1] = __float_as_uint(n[cl]); key[i - lo][2] = 0u; key[i - lo][3] = 0u; }
            else { key[i - lo][0] = canon(n[0]); key[i - lo][1] = canon(n[1]); key[i - lo][2] = canon(n[2]); key[i - lo][3] = canon(k); }
        }
        hdr[8u + c] = __float_as_uint(gmax); hdr[12u + c] = __float_as_uint(hmax);
        uint32_t packed = 0u;
        for (uint32_t cl = 0; cl < 4u; ++cl) {
            const uint32_t per = cl < 3u ? 4u : 2u, words = cl < 3u ? 4u : 8u;   // entries per group, words per entry
            uint32_t n_ent = 0;
            uint32_t* base = ent + 16u * groups;
            uint32_t done = 0u;   // records of this class already in an entry
            for (uint32_t i = 0; i < hi - lo; ++i) {
                if (cls[i] != cl || (done >> i & 1u)) continue;
                uint32_t mask = 0u, back = 0u;
                // the same plane; the same plane reversed: every key word that is a float negated (zeros stay zeros; an axis plane keeps its coordinate)
                for (uint32_t j = i; j < hi - lo; ++j) {
                    if (cls[j] != cl || (done >> j & 1u)) continue;
                    bool same = true, rev = true;
                    for (uint32_t w = 0; w < 4u; ++w) {
                        const uint32_t a = key[i][w], b = key[j][w];
                        same = same && a == b;
                        const bool coord = cl < 3u && w == 0u;   // q: not a signed quantity of the normal
                        rev = rev && (coord || a == 0u ? a == b : (a ^ 0x80000000u) == b);
                    }
                    if (same) { mask |= 0x80000000u >> j; done |= 1u << j; }
                    else if (rev) { back |= 0x80000000u >> j; done |= 1u << j; }
                }
                uint32_t* e = base + words * n_ent;
                for (uint32_t w = 0; w < words; ++w) e[w] = 0u;
                if (cl < 3u) { e[0] = key[i][0]; e[1] = key[i][1]; e[2] = mask; e[3] = back; }
                else { e[0] = key[i][0]; e[1] = key[i][1]; e[2] = key[i][2]; e[3] = key[i][3]; e[4] = mask; e[5] = back; }
                ++n_ent;
            }
            const uint32_t g = (n_ent + per - 1u) / per;
            for (uint32_t w = words * n_ent; w < 16u * g; ++w) base[w] = 0u;   // padding entries: both masks zero
            packed |= g << (8u * cl);
            groups += g;
        }
        hdr[c] = packed;   // groups per class: x | y << 8 | z << 16 | general << 24
    }
}

// Cold per-ray state parked in LDS instead of registers: the accumulator (touched once per shading event) and the
// path attenuation (once per shade).  Seven dwords per lane = 7 KB per 256-thread block, [word][lane] so a wave's
// access is one conflict-free row.  It buys the register allocator seven VGPRs on a kernel that is held at
// 6 waves/SIMD by an 80-register budget.
#ifndef PT_PARK_LDS
#define PT_PARK_LDS 1
#endif
#ifndef PT_PARK_PN
#define PT_PARK_PN 0   // A/B: parking p and n as well is slower (199.9 vs 196.2 ms): the reloads sit on the critical path
#endif
#ifndef PT_PARK_PN_GRIDS
#define PT_PARK_PN_GRIDS 1   // the same for the grid kernels only: their walks hold far more state (38 -> 22 spilled VGPRs; cornell_teapot3 859 -> 880
                             // Msamples/s, own_gems 2399 -> 2541)
#endif
#define PT_PARK_PN_FOR(GRIDS) (PT_PARK_PN || (PT_PARK_PN_GRIDS && (GRIDS) != 0))
#define PT_PARK_WORDS(GRIDS) (PT_PARK_PN_FOR(GRIDS) ? 13 : 7)
// STRIDE: words between the rows of one lane's record ([word][lane] rows of 256 lanes in k_fusedPass).  ATTE_LDS: the attenuation is
// parked too (else it stays in registers and the p / n rows move up).
template <int STRIDE, bool ATTE_LDS>
struct ParkT {
    static constexpr bool kAtteLds = ATTE_LDS;
    static constexpr int kP = ATTE_LDS ? 7 : 4, kN = kP + 3;
    float* base;   // the lane's word 0
    PT_DEV void put(int w, float v) const { base[w * STRIDE] = v; }
    PT_DEV float get(int w) const { return base[w * STRIDE]; }
    PT_DEV void acc_add(float x, float y, float z) const {
        put(0, get(0) + x); put(1, get(1) + y); put(2, get(2) + z); put(3, get(3) + 1.0f);
    }
    PT_DEV void put_pn(const Poi& q) const { put(kP, q.p.x); put(kP + 1, q.p.y); put(kP + 2, q.p.z); put(kN, q.n.x); put(kN + 1, q.n.y); put(kN + 2, q.n.z); }
    PT_DEV void get_pn(Poi& q) const { q.p = mk3(get(kP), get(kP + 1), get(kP + 2)); q.n = mk3(get(kN), get(kN + 1), get(kN + 2)); }
};
typedef ParkT<256, true> Park;

PT_DEV Box set_box(const GridArgs& S) { return set_box_of(S); }


// closest hit over every set in upload order, z-buffered through ray.maxt
// (A10 code.cl:675-800, 802-935, 937-1070; order A10 code.js:1809-1813)
template <bool FAST, int GRIDS, class PARK>
PT_DEV void closest_all(const FusedArgs& A, Ray& ray, Poi& poi, const PARK& park, bool& defer) {
    if (FAST && !(ray.mint == ray.maxt)) defer = defer || !ray_guard(ray);   // a dead ray divides nothing
    const RayRcp rr = ray_rcp<FAST>(ray);
    for (uint32_t s = 0; s < A.n_sets; ++s) {
        const GridArgs& S = A.sets[s];
        const bool live = !(ray.mint == ray.maxt);
        Hit ch;
        ch.idx = UINT32_MAX;
        if (!GRIDS || S.n == 1u) {
            if (live) {
                pt_count(PC_BOX_TESTS); pt_count(PC_BOX_LANES, true);
                const BoxHit bh = inter_aabb_t<FAST, !PT_AABB_UNSIGNED_ZERO>(ray, rr, set_box(S));
                if (bh.v) ch = (S.kind == KIND_SPHERES) ? trace_cell1<SPHERES, false, TRI_A10, FAST>(ray, bh, S) : trace_cell1<TRIANGLES, false, TRI_A10, FAST, false, PT_LANE_LISTS_FOR(FAST, GRIDS)>(ray, bh, S);
            }
        } else if (S.kind == KIND_TRIANGLES) {   // every lane of the wave enters: the tests of the walk are shared (pt_trace_coop.hpp)
            BoxHit bh = {};
            if (live) bh = inter_aabb_t<FAST, true>(ray, rr, set_box(S));
            ch = trace_dda_coop<COOP_CLOSEST, FAST, GRIDS == 1>(live && bh.v, ray, rr, bh, S, defer);
        } else if (live) {
            const BoxHit bh = inter_aabb_t<FAST, true>(ray, rr, set_box(S));
            if (bh.v) ch = trace_dda<SPHERES, false, TRI_A10, FAST, GRIDS == 1>(ray, bh, S, defer);
        }
        if (ch.idx == UINT32_MAX) continue;
        ray.maxt = ch.t;
        poi.p = fma3(ch.t, ray.d, ray.o);   // getPoint, code.cl:87
        if (S.kind == KIND_SPHERES) {
            poi.n = norm3(sub3(poi.p, ld3(((const float4*)S.prims)[ch.idx])));
            poi.matId = (int32_t)((const uint32_t*)S.matid)[ch.idx];
        } else {
            float w = 1.0f - ch.beta - ch.gamma;  // code.cl:409-411
            if (PT_LANE_LISTS_FOR(FAST, GRIDS) && S.n == 1u && S.lds_off != kNoLds) {
                // a set staged for the candidate loops carries its vertex normals and material ids in LDS too: three ds_read_b128 and a
                // ds_read_b32 instead of four dependent global loads between the hit and the bounce
                const uint32_t base = S.lds_off + 12u * S.nslots;
                const float4* nn = (const float4*)&pt_lds_dyn[base + __umul24(ch.idx, 12u)];
                poi.n = norm3(fma3(ch.gamma, ld3(nn[2]), fma3(w, ld3(nn[0]), scl3(ch.beta, ld3(nn[1])))));
                poi.matId = (int32_t)pt_lds_dyn[base + 12u * S.nslots + ch.idx];
            } else {
                const float4* nn = (const float4*)S.normals + 3u * (size_t)ch.idx;
                poi.n = norm3(fma3(ch.gamma, ld3(nn[2]), fma3(w, ld3(nn[0]), scl3(ch.beta, ld3(nn[1])))));
                poi.matId = (int32_t)(S.matid ? ((const uint32_t*)S.matid)[ch.idx] : S.mesh_matid);
            }
        }
#if PT_PARK_LDS
        if (PT_PARK_PN_FOR(GRIDS)) park.put_pn(poi);
#endif
    }
}

// per light: shadow ray, any-hit over every set, shade (A10 code.js:1817-1826; code.cl:631-673,
// 1073-1321, 1323-1364)
template <bool FAST, int GRIDS, class PARK>
PT_DEV void direct_all(const FusedArgs& A, Poi& poi, int32_t& seed, float4& acc, const PARK& park, bool& defer) {
    const float4* material = (const float4*)A.material;
    for (uint32_t l = 0; l < A.n_lights; ++l) {
        const LightArgs& L = A.lights[l];
        const bool path = poi.matId >= 0;  // initShadowTrace: a dead path draws nothing (code.cl:645-650)
        bool dark = false;
        Ray sh;
        sh.o = mk3(0.0f, 0.0f, 0.0f);
        sh.d = mk3(0.0f, 0.0f, 0.0f);
        sh.mint = PT_INF;
        sh.maxt = PT_INF;
        if (path) {
#if PT_PARK_LDS
            if (PT_PARK_PN_FOR(GRIDS)) park.get_pn(poi);
#endif
            sh = shadow_ray(poi, ld3(L.shadow), ld3(L.shadow + 3), ld3(L.shadow + 6), L.shadow[9], seed);
#if PT_SKIP_DARK_SHADOWS
            if (GRIDS) {
                // A vertex the light cannot light -- cosx * cosy == 0 in sceneRender's term (code.cl:1339-1349): the surface or the
                // emitter faces away -- adds area * (0 / r^2) * E = 0 whether or not the shadow ray is blocked (a blocked one adds the
                // literal 0; the accumulator, a sum of non-negative terms from +0, cannot tell one zero from another).  Nothing else of the
                // shadow ray survives the kernel, so it is not traced.  Needs r^2 > 0 (else 0 / 0) and finite area and irradiance.
                const float cosx = cl_clamp(dot3(sh.d, poi.n), 0.0f, 1.0f);
                const float cosy = cl_clamp(dot3(neg3(sh.d), ld3(L.scene + 3)), 0.0f, 1.0f);
                const float r = len3(sub3(poi.p, ld3(L.scene)));
                const float fin = L.scene[9] * 0.0f + L.scene[6] * 0.0f + L.scene[7] * 0.0f + L.scene[8] * 0.0f;   // 0 iff all four are finite
                dark = (cosx * cosy == 0.0f) & (r * r > 0.0f) & (fin == 0.0f);
            }
            if (FAST && !dark) defer = defer || !ray_guard(sh);
#else
            if (FAST) defer = defer || !ray_guard(sh);
#endif
        }
        const RayRcp rr = ray_rcp<FAST>(sh);
        for (uint32_t s = 0; s < A.n_sets; ++s) {
            const GridArgs& S = A.sets[s];
            const bool live = path && !dark && !(sh.mint == sh.maxt);
            Hit ch;
            bool walked = false;
            if (!GRIDS || S.n == 1u) {
                if (live) {
                    pt_count(PC_BOX_TESTS); pt_count(PC_BOX_LANES, true);
                    const BoxHit bh = inter_aabb_t<FAST, !PT_AABB_UNSIGNED_ZERO>(sh, rr, set_box(S));
                    if (bh.v) {
                        ch = (S.kind == KIND_SPHERES) ? trace_cell1<SPHERES, true, TRI_A10, FAST, true>(sh, bh, S) : trace_cell1<TRIANGLES, true, TRI_A10, FAST, true, PT_LANE_LISTS_FOR(FAST, GRIDS)>(sh, bh, S);
                        walked = true;
                    }
                }
            } else if (S.kind == KIND_TRIANGLES) {
                BoxHit bh = {};
                if (live) bh = inter_aabb_t<FAST, true>(sh, rr, set_box(S));
                walked = live && bh.v;
                ch = trace_dda_coop<COOP_ANY, FAST, GRIDS == 1>(walked, sh, rr, bh, S, defer);
            } else if (live) {
                const BoxHit bh = inter_aabb_t<FAST, true>(sh, rr, set_box(S));
                if (bh.v) {
                    ch = trace_dda<SPHERES, true, TRI_A10, FAST, GRIDS == 1>(sh, bh, S, defer);
                    walked = true;
                }
            }
            if (!walked) continue;
            sh.maxt = ch.t;
            if (ch.idx != UINT32_MAX) sh.mint = ch.t;
        }
        if (!path || (uint32_t)poi.matId >= A.nmat) continue;  // out-of-range id: shade nothing (see k_sceneRender)
        pt_count(PC_SHADE); pt_count(PC_SHADE_LANES, true);
        float4 c4 = material[poi.matId];
#if PT_PARK_LDS
        if (PT_PARK_PN_FOR(GRIDS)) park.get_pn(poi);
        if (PARK::kAtteLds) poi.atte = mk3(park.get(4), park.get(5), park.get(6));
        f3 c = shade_vertex(poi, sh, mk3(c4.x, c4.y, c4.z), ld3(L.scene), ld3(L.scene + 3), ld3(L.scene + 6), L.scene[9]);
        if (PARK::kAtteLds) { park.put(4, poi.atte.x); park.put(5, poi.atte.y); park.put(6, poi.atte.z); }
        park.acc_add(c.x, c.y, c.z);
#else
        f3 c = shade_vertex(poi, sh, mk3(c4.x, c4.y, c4.z), ld3(L.scene), ld3(L.scene + 3), ld3(L.scene + 6), L.scene[9]);
        acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
#endif
    }
}

// Block prologue of the fused kernels.  GRIDS: the cell-offset tables of the grid sets (uint[n^3 + 1] each) are copied into LDS once
// per block, before any thread leaves: launch_fused gave every set that fits a slot (GridArgs::lds_off).  The primitives of a grid stay
// in memory.  PT_LANE_LISTS: likewise the prepared records of the single-cell triangle sets launch_fused gave a slot (the candidate
// loops fetch them per lane by ds_read_b128).
// The k x k lens grid's coordinates (code.cl:482-509): coord = delta / 2 and then `+= delta` per step -- sample (i, j) of every pixel needs
// the i-th and the j-th partial sum of that chain.  One thread walks the chain once per block and leaves the k values in LDS (as
// every lane walking it to its own i and j it cost up to 2 (k - 1) dependent additions per sample: 30 of them at 256 rays per pixel, 62 at
// 1024).  k > kLensTab: the lanes walk.
constexpr uint32_t kLensTab = 64;
static_assert(kLensTab % 4u == 0u, "the static LDS ahead of pt_lds_dyn stays a multiple of 16 bytes");
__shared__ float pt_lens_tab[kLensTab];
__shared__ uint32_t pt_blk_defer[4];   // in-pass resolve: "a sample of this block left the guard windows" (word 0; four words keep what follows 16-byte aligned)
// (the ray count is laundered through an SGPR: as a common subexpression of the block prologue and of every sample's set-up, the float made from it
// stayed alive in a VGPR between the two -- the one register the 64-register build had to spill)
PT_DEV uint32_t lens_side(const FusedArgs& A) {
    uint32_t rpp = A.rpp;
    asm volatile("" : "+s"(rpp));
    return f2u_uniform(cl_sqrt((float)rpp));
}

template <bool FAST, int GRIDS>
PT_DEV void stage_block(const FusedArgs& A) {
    if (threadIdx.x == 0u) pt_blk_defer[0] = 0u;
    if (A.rpp > 1u && threadIdx.x == 0u) {
        const uint32_t side = lens_side(A);
        if (side <= kLensTab) {
            const float delta = 1.0f / (float)side;
            float c = delta / 2.0f;
            for (uint32_t k = 0; k < side; ++k) { pt_lens_tab[k] = c; c += delta; }
        }
    }
    if (!(GRIDS == 1 || PT_LANE_LISTS_FOR(FAST, GRIDS))) __syncthreads();
    if (GRIDS == 1 || PT_LANE_LISTS_FOR(FAST, GRIDS)) {
        for (uint32_t s = 0; s < A.n_sets; ++s) {
            const GridArgs& S = A.sets[s];
            if (S.lds_off == kNoLds) continue;
            if (S.n == 1u) {
                if (!PT_LANE_LISTS_FOR(FAST, GRIDS)) continue;
                // [records 12 words each][vertex normals 12 words each][material ids, one word each: a mesh's single id repeated]
                const uint32_t words = S.nslots * 12u;
                const uint32_t* src = (const uint32_t*)S.prims;
                const uint32_t* nrm = (const uint32_t*)S.normals;
                const uint32_t* mid = (const uint32_t*)S.matid;
                for (uint32_t k = threadIdx.x; k < words; k += 256u) { pt_lds_dyn[S.lds_off + k] = src[k]; pt_lds_dyn[S.lds_off + words + k] = nrm[k]; }
                for (uint32_t k = threadIdx.x; k < S.nslots; k += 256u) pt_lds_dyn[S.lds_off + 2u * words + k] = mid ? mid[k] : S.mesh_matid;
            } else if (GRIDS == 1) {
                const uint32_t words = S.n * S.n * S.n + 1u;
                const uint32_t* src = (const uint32_t*)S.off;
                for (uint32_t k = threadIdx.x; k < words; k += 256u) pt_lds_dyn[S.lds_off + k] = src[k];
            }
        }
        __syncthreads();
    }
}

// initTrace (code.cl:458-543) for one ray id of the tile: thin-lens ray through its pixel, clipped to the scene box
// (tile-local ray ids fit 32 bits: mirt_render_pass refuses a tile of more than 2^32 - 256 rays)
PT_DEV Ray primary_ray(const FusedArgs& A, uint32_t lid) {
    // (the host makes rays_per_pixel k x k; when k is a power of two -- 1, 4, 16, 64, 256, 1024 -- the pixel and the sample are a shift and a mask)
    const bool pow2 = (A.rpp & (A.rpp - 1u)) == 0u;   // wave-uniform
    const uint32_t lpix = pow2 ? lid >> (uint32_t)__builtin_ctz(A.rpp) : lid / A.rpp;
    const uint32_t smp = pow2 ? lid & (A.rpp - 1u) : lid - lpix * A.rpp;
    const uint32_t lrow = lpix / A.width;
    const uint32_t col = lpix - lrow * A.width;
    const uint32_t row = A.row0 + lrow;
    Cam cam;
    cam.eye = ld3(A.cam); cam.U = ld3(A.cam + 3); cam.V = ld3(A.cam + 6); cam.W = ld3(A.cam + 9);
    cam.width = A.cam[12]; cam.height = A.cam[13];
    cam.cols = f2u_uniform(A.cam[14]); cam.rows = f2u_uniform(A.cam[15]);
    Box bound;
    bound.lo = mk3(A.bound[0], A.bound[1], A.bound[2]);
    bound.hi = mk3(A.bound[4], A.bound[5], A.bound[6]);
    const f3 fp = focal_point(cam, (float)col, (float)row, A.focal_length);
    float cx, cy;
    if (A.rpp > 1) {
        // un-jittered k x k lens grid; coordinates accumulate by repeated addition in the
        // reference (coord += delta), so they are rebuilt the same way
        const uint32_t side = lens_side(A);
        const uint32_t i = smp / side, j = smp - i * side;
        if (side <= kLensTab) {   // the chain's partial sums, left in LDS by stage_block
            cy = pt_lens_tab[i];
            cx = pt_lens_tab[j];
        } else {
            const float delta = 1.0f / (float)side;
            cy = delta / 2.0f;
            for (uint32_t k = 0; k < i; ++k) cy += delta;
            cx = delta / 2.0f;
            for (uint32_t k = 0; k < j; ++k) cx += delta;
        }
    } else {
        float2 c = ((const float2*)A.uv)[lpix];
        cx = c.x;
        cy = c.y;
    }
    Ray ray = thin_lens_ray(cam, fp, A.lens_rad, cx, cy);
    clip_to(ray, bound);
    return ray;
}

// copyToPixel inside the pass (A10 code.cl:1366-1386) for the block that holds ray ids [first, first + 256): `rows` = the block's parked
// accumulators, [channel][lane] (rows 0..3 of the park area), final for every lane (the caller's barrier).  rpp divides 256, so the block
// holds 256 / rpp whole pixels; one lane per (pixel, channel) adds that pixel's rpp samples in the reference's order -- sequential in i,
// from +0: the fp32 sum is order-dependent -- reading them four at a time (a channel's row is contiguous in LDS); the four lanes of a pixel
// (one DPP quad) hand their sums to the first, which writes `radiance` (the sums) and `pixel` (the tone-scaled RGBA8, truncating).
#ifndef PT_RESOLVE_PRIO
#define PT_RESOLVE_PRIO 1
#endif
// The block of 256 consecutive ray ids a workgroup renders (outside the redo loop): its own number -- or, a pixel of more than 256 rays being resolved
// in `chunks` launches, block `chunk` of pixel blockIdx.x (FusedArgs::chunks).  Scalar arithmetic on kernel arguments, re-derived where it is used.
PT_DEV uint32_t wg_block(const FusedArgs& A) { return blockIdx.x * A.chunks + A.chunk; }
PT_DEV void resolve_block(const FusedArgs& A, const float* rows, uint64_t first, uint64_t n_local) {
    const uint32_t rpp = A.rpp, per = rpp < 256u ? rpp : 256u, ppb = 256u / per;   // per: the rays of ONE pixel this block holds (rpp > 256: FusedArgs::chunks)
    const uint32_t pix0 = (uint32_t)(first / rpp), npix = (uint32_t)(n_local / rpp);
#if PT_RESOLVE_PRIO
    // The sums are chains of dependent additions (256 long at 256 rays per pixel, on four lanes) at the very end of a block whose other waves
    // have left: until the chain ends the block's LDS and this wave's slot are held.  At the top priority the chain's instructions issue as they
    // become ready instead of waiting their turn among the SIMD's other waves.
    __builtin_amdgcn_s_setprio(3);
#endif
    for (uint32_t q = threadIdx.x; q < 4u * ppb; q += 256u) {   // whole quads: 4 ppb is a multiple of 4, and so is every q - threadIdx.x
        const uint32_t j = q >> 2, c = q & 3u;
        const float* r = rows + c * 256u + j * per;
        float s = 0.0f;
        // a later block of a pixel of more than 256 rays: the chain goes on from the sum over the blocks before it (FusedArgs::chunks)
        if (A.chunk != 0u && pix0 + j < npix) s = ((const float*)A.radiance)[4u * (size_t)(pix0 + j) + c];
        if (per >= 4u) {
            const float4* r4 = (const float4*)r;   // 16-byte aligned: the rows are, and `per` is a multiple of 4
            for (uint32_t i = 0; i < per / 4u; ++i) { const float4 v = r4[i]; s += v.x; s += v.y; s += v.z; s += v.w; }
        } else {
            for (uint32_t i = 0; i < per; ++i) s += r[i];
        }
        const int si = (int)__float_as_uint(s);
        const float x = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp(si, 0x00, 0xf, 0xf, true));   // quad_perm [0,0,0,0]
        const float y = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp(si, 0x55, 0xf, 0xf, true));   // [1,1,1,1]
        const float z = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp(si, 0xaa, 0xf, 0xf, true));   // [2,2,2,2]
        const float w = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp(si, 0xff, 0xf, 0xf, true));   // [3,3,3,3]
        if (c != 0u || pix0 + j >= npix) continue;
        if (A.radiance) ((float4*)A.radiance)[pix0 + j] = make_float4(x, y, z, w);
        if (A.pixel) {
            const float sc = 255.0f * A.res_m;
            const float cx = cl_clamp((x * sc) * 1.8f, 0.0f, 255.0f), cy = cl_clamp((y * sc) * 1.8f, 0.0f, 255.0f), cz = cl_clamp((z * sc) * 1.8f, 0.0f, 255.0f);
            ((uchar4*)A.pixel)[pix0 + j] = make_uchar4((unsigned char)f2u(cx), (unsigned char)f2u(cy), (unsigned char)f2u(cz), 255);
        }
    }
}

#ifndef PT_FUSED_WAVES
#define PT_FUSED_WAVES 6   // waves per SIMD the register allocator must leave room for (A/B without SLP packing: 5 -> 182.8 ms, 6 -> 178.3, 7 -> 181.0, 8 -> 192.1)
#endif
// FAST = true : the optimistic kernel.  Every division in the traversal is one of the exact cheap forms; a sample whose rays
//               ever leave the guard window sets its bit in `defer_mask` and leaves seeds[]/acu[] untouched.
// FAST = false: the exact kernel (true divisions).  With `list` it recomputes the deferred samples; with list == nullptr it
//               is the whole pass (geometry outside the guard, or PT_EXACT_FAST_DIV = 0).
#ifndef PT_FUSED_WAVES_FAST
#define PT_FUSED_WAVES_FAST 8   // the optimistic kernel without the grid walk: 64 VGPRs, no scratch, since its body is straight-line (round 3: 7 -> 109.4 ms, 8 -> 107.7;
                                // round 1, same box: 5 -> 216.7 ms, 6 -> 216.6, 7 -> 213.4, 8 -> 213.7 at depth 8)
#endif
// GRIDS = 0    : every set is a single cell (n == 1: the reference's loose spheres and triangles, A10 code.js:399): only the
//                wave-uniform loops are compiled in.  Without the DDA the register allocator needs 72 VGPRs and no scratch
//                (80 + 56 B with it): 157.3 -> 152.6 ms on the headline scene.  launch_fused picks it when every set has n == 1.
// GRIDS = 1, 2 : sets with n > 1 walk their grid per lane (trace_dda); 1: every cell-offset table is staged in LDS, 2: none is
//                (a scene whose tables exceed kLdsOffWords).
// Tried and dropped for GRIDS: packing the rays that hit a mesh's box across the block's four waves through LDS (one wave walks
// 64 packed rays, three wait at a barrier).  It cuts VALU instructions 4x on those walks and was 30 % SLOWER (cornell_teapot3
// 1080p x16: 72.2 -> 93.8 ms): the waiting waves keep their registers, so each SIMD is left with too few runnable waves.
// What pays instead is sharing the TESTS inside each wave, no barrier, no idle wave: pt_trace_coop.hpp (65.4 -> 40.2 ms).
#ifndef PT_STAGE_TABLES
#define PT_STAGE_TABLES 1
#endif
#ifndef PT_FUSED_WAVES_GRIDS
#define PT_FUSED_WAVES_GRIDS 6   // the grid walk is latency-bound: it wants waves.  Round 3, cornell_teapot3 1080p x 16: 4 waves per SIMD 38.2 ms, 5 (96 VGPRs, no
                                // scratch) 31.6, 6 (80 VGPRs, 18 spilled around the walks, 64 B of scratch) 29.3 -- 6 needs a block's LDS within 26 880 B (six
                                // blocks per CU at the 1280-byte granule), which it is since the owner's ray travels by ds_bpermute (pt_trace_coop.hpp)
                                // (round 2, at five blocks per CU whatever this said: 6 -> 48.8 ms, 5 -> 43.0, 4 -> 48.1)
#endif
// WAVES != 0: an occupancy variant of the optimistic grid kernels.  PT_FUSED_WAVES_GRIDS waves per SIMD only exist while a block's LDS lets as
// many blocks share a CU (26 880 B at six); a scene with bigger cell tables gets five blocks at best, and for it the 96-register build --
// no scratch -- is the better kernel: launch_fused picks by the bytes it is about to ask for.
template <bool FAST, int GRIDS, int WAVES = 0>
__global__ void __launch_bounds__(256, WAVES ? WAVES : (GRIDS ? PT_FUSED_WAVES_GRIDS : (FAST ? PT_FUSED_WAVES_FAST : PT_FUSED_WAVES))) k_fusedPass(const FusedArgs A, uint32_t* defer_mask, const uint32_t* redo_mask, uint32_t redo_words) {
    const uint64_t n_local = (uint64_t)A.nrows * A.width * A.rpp;
    stage_block<FAST, GRIDS>(A);
    // Exact kernel in redo mode (`redo_mask`: the bits the optimistic kernel set): one thread per 32-sample word, a loop over its
    // set bits -- no list, no count, no host round trip between the two kernels.  Otherwise: one thread, one sample, one trip.
    // GRIDS: the walk shares its triangle tests across the wave (pt_trace_coop.hpp), so every lane stays in to the end: a lane
    // without a sample of its own (past the end of the tile; no bit left in its redo word) rides along on the tile's last sample
    // and writes nothing.
    // In-pass resolve (A.resolve): the unit of everything is the BLOCK of 256 consecutive ray ids -- the optimistic kernel hands a whole block
    // over when one of its samples left the guard windows (a bit per block in `defer_mask`; nothing of the block is written), and the exact
    // kernel's redo mode is one block per 32-block word of that mask, every thread one sample of each marked block in turn.  Every thread stays
    // in to the block's barrier: a lane past the end of the tile rides along as in the grid kernels.
    uint64_t base = (uint64_t)wg_block(A) * 256u + threadIdx.x;   // (every launch of this kernel uses 256-thread blocks: launch_fused)
    uint32_t todo = 1u;
    uint32_t stride = 1u;
    if (!FAST && redo_mask) {
        if (A.resolve) {
            todo = redo_mask[blockIdx.x] & A.chunk_bits;   // (the grid is one block per word; the bits of this launch's blocks: FusedArgs::chunks)
            base = (uint64_t)blockIdx.x * (32u * 256u) + threadIdx.x;
            stride = 256u;
        } else {
            if (GRIDS) todo = base < redo_words ? redo_mask[base] : 0u;
            else {
                if (base >= redo_words) return;
                todo = redo_mask[base];
            }
            base *= 32u;
        }
    }
  for (;; todo &= todo - 1u) {
    bool valid = todo != 0u;
    if (GRIDS) { if (__builtin_amdgcn_ballot_w64(valid) == 0ull) break; }
    else if (!valid) break;
    // the id is checked in 64 bits (the last block of a tile of nearly 2^32 rays reaches past it) and kept in 32
    const uint64_t lid64 = base + (valid ? (uint32_t)__builtin_ctz(todo) * stride : 0u);
    uint32_t lid = (uint32_t)lid64;
    if (lid64 >= n_local) {
        if (!GRIDS && !A.resolve) return;
        valid = false;
        lid = (uint32_t)(n_local - 1u);
    }
    bool defer = false;
    int32_t seed = A.seeds[lid];
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // initAcu (A10 code.cl:448-456) when the pass is a frame's first
    if (!A.fresh) acc = ((const float4*)A.acu)[lid];
    Park park;
#if PT_PARK_LDS
    __shared__ __attribute__((aligned(16))) float park_mem[PT_PARK_WORDS(GRIDS)][256];
    park.base = &park_mem[0][threadIdx.x];
    park.put(0, acc.x); park.put(1, acc.y); park.put(2, acc.z); park.put(3, acc.w);
    park.put(4, 1.0f); park.put(5, 1.0f); park.put(6, 1.0f);
#endif
    Ray ray = primary_ray(A, lid);
    Poi poi;
    poi.p = mk3(0.0f, 0.0f, 0.0f);
    poi.n = mk3(0.0f, 0.0f, 0.0f);
    poi.atte = mk3(1.0f, 1.0f, 1.0f);
    poi.matId = -1;

    // segment 0 is the primary ray; segments 1..bounces start with bouncePaths (code.js:1829-1846)
    pt_count(PC_WAVES);
    for (uint32_t seg = 0; seg <= A.bounces; ++seg) {
        pt_count(PC_SEGMENTS);
        if (seg > 0) {
            if (poi.matId >= 0) {
                pt_count(PC_BOUNCE); pt_count(PC_BOUNCE_LANES, true);
#if PT_PARK_LDS
                if (PT_PARK_PN_FOR(GRIDS)) park.get_pn(poi);
#endif
                ray = bounce_ray(poi, seed);
            } else {
                ray.mint = PT_INF;
                ray.maxt = PT_INF;
            }
        }
        if (PT_COUNT && !(ray.mint == ray.maxt)) pt_count(PC_SEG_LANES, true);
        closest_all<FAST, GRIDS, Park>(A, ray, poi, park, defer);
        if (seg == 0) {
            for (uint32_t l = 0; l < A.n_lights; ++l) {  // lightRender (code.cl:600-629), primary segment only
                if (ray.mint == ray.maxt) continue;
                const LightArgs& L = A.lights[l];
                f3 irr = norm3(ld3(L.light + 6));
                if (!light_visible(ray, ld3(L.light), ld3(L.light + 3), L.light[9])) continue;
                ray.mint = PT_INF;
                ray.maxt = PT_INF;
                poi.matId = -1;
#if PT_PARK_LDS
                park.acc_add(irr.x, irr.y, irr.z);
#else
                acc.x += irr.x; acc.y += irr.y; acc.z += irr.z; acc.w += 1.0f;
#endif
            }
        }
        direct_all<FAST, GRIDS, Park>(A, poi, seed, acc, park, defer);
    }

    if (FAST) {
        // the ray id again, from the thread index (one sample per thread in this kernel): re-deriving it here costs a few instructions,
        // keeping it (and the 64-bit addresses made from it) alive across the whole path cost spilled registers in every variant
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const uint64_t again = (uint64_t)wg_block(A) * 256u + t;
        lid = (uint32_t)again;
        if (GRIDS || A.resolve) valid = again < n_local;   // (a lane past the end of the tile rode along on the tile's last sample)
    }
    if (A.resolve) {   // wave-uniform (a kernel argument)
#if !PT_PARK_LDS
        __shared__ __attribute__((aligned(16))) float park_mem[4][256];
        park_mem[0][threadIdx.x] = acc.x; park_mem[1][threadIdx.x] = acc.y; park_mem[2][threadIdx.x] = acc.z; park_mem[3][threadIdx.x] = acc.w;
#endif
        if (FAST && defer) pt_blk_defer[0] = 1u;
        __syncthreads();   // every lane's accumulator is final in LDS, and so is the flag
        if (FAST && pt_blk_defer[0] != 0u) {   // the whole block goes to the exact kernel: its seeds stay as they were, no pixel of it is written
            if (threadIdx.x == 0u) atomicOr(&defer_mask[wg_block(A) >> 5], 1u << (wg_block(A) & 31u));
            return;
        }
        if (valid) {
            A.seeds[lid] = seed;
            if (A.acu) ((float4*)A.acu)[lid] = make_float4(park_mem[0][threadIdx.x], park_mem[1][threadIdx.x], park_mem[2][threadIdx.x], park_mem[3][threadIdx.x]);
        }
        // the block's first ray id: wave-uniform (blockIdx alone in the optimistic kernel; the marked block of this trip in the redo loop)
        const uint64_t first = FAST || stride != 256u ? (uint64_t)wg_block(A) * 256u
                                                      : (uint64_t)blockIdx.x * (32u * 256u) + (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctz(todo)) * 256u;
        resolve_block(A, &park_mem[0][0], first, n_local);
        if (FAST) return;
        __syncthreads();   // the redo loop's next block parks into the same rows
        continue;
    }
    // The optimistic kernel has one sample per thread: it LEAVES here, so the compiler sees a straight-line body and not a loop (the
    // exact kernel's redo mode does loop over the set bits of its word).  As a loop -- its exit in the grid kernels is a wave ballot, opaque
    // to the compiler -- every sample-independent value of the body (the camera set-up, sqrt(rpp), the lens-grid reciprocals ...) was
    // hoisted out and kept alive through the whole path: that was what the grid kernels spilled.
    if (FAST && defer) {   // hand the sample to the exact kernel: its inputs stay as they were
        if (valid) atomicOr(&defer_mask[lid >> 5], 1u << (lid & 31u));
        return;
    }
    if (valid) {
        A.seeds[lid] = seed;
#if PT_PARK_LDS
        acc = make_float4(park.get(0), park.get(1), park.get(2), park.get(3));
#endif
        ((float4*)A.acu)[lid] = acc;
    }
    if (FAST) return;
  }
}


// deferred-sample bookkeeping: count the set bits (only when the host asks, mirt_pass_deferred)
__global__ void __launch_bounds__(256) k_deferCount(const uint32_t* mask, uint32_t words, uint32_t* count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = (i < words) ? (uint32_t)__builtin_popcount(mask[i]) : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
void launch_fused(hipStream_t s, const FusedArgs& a, bool fast, uint32_t* defer_mask, const uint32_t* redo_mask, uint32_t redo_words) {
    const uint64_t n = redo_mask ? redo_words : (uint64_t)a.nrows * a.width * a.rpp;
    if (!n) return;
    bool grids = false;
    for (uint32_t i = 0; i < a.n_sets; ++i) grids = grids || a.sets[i].n != 1u;
    FusedArgs b = a;   // LDS slots for the cell-offset tables: all of them or none (the walk's table reads are compiled for one address space)
    uint64_t used = 0;
    for (uint32_t i = 0; i < b.n_sets; ++i) {
        b.sets[i].lds_off = kNoLds;
        if (b.sets[i].n > 1u) { b.sets[i].lds_off = kCoopWordsPerBlock + (uint32_t)(used < kLdsOffWords ? used : kLdsOffWords); used += (uint64_t)b.sets[i].n * b.sets[i].n * b.sets[i].n + 1u; }
    }
    const bool staged = PT_STAGE_TABLES && used <= kLdsOffWords;
    if (grids && !staged)
        for (uint32_t i = 0; i < b.n_sets; ++i) b.sets[i].lds_off = kNoLds;
    // ... then the prepared records of the single-cell triangle sets, for the candidate loops of the optimistic kernel: all that fit
    // kLdsTriMax records, in upload order (a set without a slot runs the wave-uniform loop)
    uint32_t tri_words = 0;
    const uint32_t tri_base = grids ? kCoopWordsPerBlock + (staged ? (uint32_t)used : 0u) : 0u;   // multiples of 4 words: kCoopWordsPerBlock is, `used` is rounded up below
    const uint32_t tri_base4 = (tri_base + 3u) & ~3u;
    if (fast && PT_LANE_LISTS_FOR(true, grids ? 1 : 0)) {
        uint32_t tris = 0;
        for (uint32_t i = 0; i < b.n_sets; ++i) {
            GridArgs& S = b.sets[i];
            if (S.n != 1u || S.kind != KIND_TRIANGLES || !S.pnorm || S.nslots == 0u || tris + S.nslots > kLdsTriMax) continue;
            S.lds_off = tri_base4 + tri_words;
            tri_words += S.nslots * 28u;   // records, vertex normals, material ids (stage_block), rounded up to whole float4s
            tris += S.nslots;
        }

    }
    // redo mode: one thread per 32-sample word of the mask, or (in-pass resolve: the mask is per block) one block per 32-block word
    // (resolving a pixel of more than 256 rays: one workgroup per pixel and launch, FusedArgs::chunks -- n is a multiple of 256 * chunks then)
    const dim3 grid(redo_mask && a.resolve ? (unsigned)redo_words : (unsigned)((n + 255) / 256 / (a.chunks ? a.chunks : 1u)));
    // dynamic LDS: the waves' exchange areas, then the staged tables (what the scene needs, not the 16 KB cap: occupancy), then the staged triangles
    const size_t lds_tri = tri_words ? (size_t)(tri_base4 - tri_base + tri_words) * 4u : 0u;
    const size_t lds2 = (size_t)kCoopWordsPerBlock * 4u + lds_tri, lds = (size_t)kCoopWordsPerBlock * 4u + (staged ? (size_t)used * 4u : 0u) + lds_tri;
    if (fast) {
        // LDS is handed out in 1280-byte granules, 128 of them per CU: the blocks per CU this launch can have, and the waves per SIMD worth compiling for
        static const int force_waves = [] { const char* e = getenv("MIRT_GRID_WAVES"); return e ? atoi(e) : 0; }();   // A/B and test switch
        const size_t fixed = sizeof(float) * ((size_t)PT_PARK_WORDS(1) * 256u + kLensTab);   // the grid kernels' static LDS: parked state, lens table
        const size_t want = grids && staged ? lds : lds2, granules = (want + fixed + 1279u) / 1280u;
        const bool five = force_waves ? force_waves == 5 : (granules ? 128u / granules : 8u) < (unsigned)PT_FUSED_WAVES_GRIDS;
        if (grids && staged && five) hipLaunchKernelGGL((k_fusedPass<true, 1, 5>), grid, dim3(256), lds, s, b, defer_mask, (const uint32_t*)nullptr, 0u);
        else if (grids && staged) hipLaunchKernelGGL((k_fusedPass<true, 1>), grid, dim3(256), lds, s, b, defer_mask, (const uint32_t*)nullptr, 0u);
        else if (grids && five) hipLaunchKernelGGL((k_fusedPass<true, 2, 5>), grid, dim3(256), lds2, s, b, defer_mask, (const uint32_t*)nullptr, 0u);
        else if (grids) hipLaunchKernelGGL((k_fusedPass<true, 2>), grid, dim3(256), lds2, s, b, defer_mask, (const uint32_t*)nullptr, 0u);
        else hipLaunchKernelGGL((k_fusedPass<true, 0>), grid, dim3(256), lds_tri, s, b, defer_mask, (const uint32_t*)nullptr, 0u);
    } else {
        if (grids && staged) hipLaunchKernelGGL((k_fusedPass<false, 1>), grid, dim3(256), lds, s, b, (uint32_t*)nullptr, redo_mask, redo_words);
        else if (grids) hipLaunchKernelGGL((k_fusedPass<false, 2>), grid, dim3(256), lds2, s, b, (uint32_t*)nullptr, redo_mask, redo_words);
        else hipLaunchKernelGGL((k_fusedPass<false, 0>), grid, dim3(256), 0, s, b, (uint32_t*)nullptr, redo_mask, redo_words);
    }
}
bool fused_fast_available() { return PT_EXACT_FAST_DIV != 0; }
#if PT_COUNT
int debug_counters(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pt_counters), sizeof(unsigned long long) * PC_COUNT) != hipSuccess) return -1;
    if (reset) { unsigned long long z[PC_COUNT] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(pt_counters), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
void launch_deferCount(hipStream_t s, const uint32_t* mask, uint32_t words, uint32_t* count) {
    if (words) hipLaunchKernelGGL(k_deferCount, dim3((words + 255) / 256), dim3(256), 0, s, mask, words, count);
}

size_t prepared_bytes(uint32_t count) { return prepared_planes_offset(count) + (count <= kLdsTriMax ? prepared_planes_bytes(count) : 0); }
void launch_prepTriangles(hipStream_t s, const void* pos, void* out, uint32_t count, uint32_t* insane_word) {
    if (!count) return;
    hipLaunchKernelGGL(k_prepTriangles, dim3((count + 255) / 256), dim3(256), 0, s, (const float4*)pos, (float4*)out, count, insane_word);
    if (count <= kLdsTriMax) hipLaunchKernelGGL(k_planeList, dim3(1), dim3(64), 0, s, (const float4*)out, count, (uint32_t*)((char*)out + prepared_planes_offset(count)));
}

}  // namespace pt
