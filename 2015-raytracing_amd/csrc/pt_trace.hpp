// pt_trace.hpp -- traversal of one primitive set over PREPARED geometry, shared by the fused pass
// (pt_kernels_fused.hip) and the single-frame kernels of Assign04/07 (pt_kernels_frame.hip).
//
// Prepared triangle = 3 x float4 {p0.xyz, n.x} {e1.xyz, n.y} {e2.xyz, n.z} with e1 = p1-p0, e2 = p2-p0,
// n = cross(e2,e1): the ray-independent head of the reference's Moeller-Trumbore (A10 code.cl:252-256),
// computed once by k_prepTriangles with the same fp32 operations.
#pragma once
#include "pt_device.hpp"
#include "pt_launch.hpp"

#ifndef PT_UNROLL_UNIFORM
#define PT_UNROLL_UNIFORM 1   // primitives per trip of the wave-uniform loops
#endif

namespace pt {

// -DPT_COUNT=1: a profiling build (profiles/trip_counts.py) -- wave-level trip counters of the fused pass's loops, so the static instruction
// count of each loop body can be weighted by how often it really runs.  One atomic per counted event from the first active lane; `lanes`
// variants add the number of active lanes.  Not a product build: the counters are read through mirt_debug_counters, which only this build exports.
#ifndef PT_COUNT
#define PT_COUNT 0
#endif
enum PtCounter { PC_WAVES = 0, PC_SEGMENTS, PC_SEG_LANES, PC_Q_CLOSEST, PC_Q_CLOSEST_LANES, PC_SWEEP_CLOSEST, PC_TRIPS_CLOSEST, PC_TRIP_LANES_CLOSEST,
                 PC_Q_SHADOW, PC_Q_SHADOW_LANES, PC_SWEEP_SHADOW, PC_TRIPS_SHADOW, PC_TRIP_LANES_SHADOW,
                 PC_SPH_Q_CLOSEST, PC_SPH_Q_CLOSEST_LANES, PC_SPH_TESTS_CLOSEST, PC_SPH_ROOTS_CLOSEST, PC_SPH_Q_SHADOW, PC_SPH_Q_SHADOW_LANES, PC_SPH_TESTS_SHADOW, PC_SPH_ROOTS_SHADOW,
                 PC_BOX_TESTS, PC_BOX_LANES, PC_SHADE, PC_SHADE_LANES, PC_BOUNCE, PC_BOUNCE_LANES,
                 // the shared-test grid walk (pt_trace_coop.hpp): walks a wave enters, lanes that want one, outer phases, wave-level empty-cell steps and the lanes
                 // taking them, pooled rounds, pairs; [closest, shadow] each
                 PC_GRID_WALKS, PC_GRID_WANT_LANES = PC_GRID_WALKS + 2, PC_GRID_PHASES = PC_GRID_WALKS + 4, PC_GRID_A_STEPS = PC_GRID_WALKS + 6,
                 PC_GRID_A_LANE_STEPS = PC_GRID_WALKS + 8, PC_GRID_ROUNDS = PC_GRID_WALKS + 10, PC_GRID_PAIRS = PC_GRID_WALKS + 12, PC_COUNT = 48 };
#if PT_COUNT
static __device__ unsigned long long pt_counters[PC_COUNT];
__device__ __forceinline__ void pt_count(int i, bool lanes = false, unsigned long long n = 1ull) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    const unsigned me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (me == (unsigned)__builtin_ctzll(m)) atomicAdd(&pt_counters[i], lanes ? (unsigned long long)__builtin_popcountll(m) : n);
}
#else
__device__ __forceinline__ void pt_count(int, bool = false, unsigned long long = 1ull) {}
#endif

struct Hit { uint32_t idx; float t, beta, gamma; };

// Geometry the kernels only read, fetched in wave-uniform loops: through the CONSTANT address space, so the loads stay scalar
// (s_load_dwordx4 / x16 into SGPRs) wherever the address is uniform.  Through a plain global pointer the back end may only use a scalar
// load when nothing in the kernel can have stored to memory before it: a kernel that writes results inside its main loop (the
// compacting variant, profiles/experiments/wave_queue_compaction.patch) silently got per-lane global_loads for every geometry read
// after the first store.  The scene buffers are never written by a pass.
#define PT_CONST_AS __attribute__((address_space(4)))
typedef float pt_v4f __attribute__((ext_vector_type(4)));
typedef float pt_v16f __attribute__((ext_vector_type(16)));
PT_DEV float4 ldc4(const void* base, size_t idx) {   // float4 number idx of a read-only array
    const pt_v4f v = ((const pt_v4f PT_CONST_AS*)base)[idx];
    return make_float4(v.x, v.y, v.z, v.w);
}
PT_DEV uint32_t ldc_u32(const void* base, size_t idx) { return ((const uint32_t PT_CONST_AS*)base)[idx]; }

// Moeller-Trumbore on a prepared triangle; same operations, same order and the same
// accept/reject predicates as the reference's interTriangle (A10 code.cl:250-288), written without early exits:
// in a wave-uniform loop the 64 lanes leave at different tests anyway.
// TRI_A10: closed t interval, no gamma > 1 test (A10 code.cl:273, 280)
// TRI_A07: open t interval, no gamma > 1 test   (A07 code.cl:188, 195)
// TRI_A04: open t interval, gamma > 1 rejects    (A04 code.cl:170, 177)
#ifndef PT_GUARD_INT
#define PT_GUARD_INT 1
#endif
enum TriRule { TRI_A10 = 0, TRI_A07 = 1, TRI_A04 = 2 };
// FAST: 1/div by the 3-operation refined reciprocal (pt_numerics.hpp), exact whenever div is zero or inside its window, which
// the guards guarantee (ray_guard() on the ray side, GridArgs::fast_ok on the geometry side).  A lane whose ray fails the guard
// still runs this code -- on values nobody will read: the optimistic kernel marks that sample `deferred` and the exact kernel
// (FAST = false, true divisions everywhere) recomputes it from its untouched seed and accumulator (pt_kernels_fused.hip).
#ifndef PT_UNIFORM_CULL
#define PT_UNIFORM_CULL 0
#endif
#ifndef PT_SPHERE_WAVE_SKIP
#define PT_SPHERE_WAVE_SKIP 1
#endif
// WAVE_CULL (the wave-uniform loops: all lanes hold the SAME triangle): when the triangle faces away from every lane's ray
// (div <= 0 in all of them: coherent primary and shadow rays against half of a closed room's walls) the rest of the test is skipped for the
// wave -- the reference's first early-out (code.cl:259), taken when the whole wave takes it.
template <int RULE = TRI_A10, bool FAST = false, bool WAVE_CULL = false>
PT_DEV bool tri_test(f3 o, f3 d, float cmin, float cmax, const float4 A, const float4 B, const float4 C,
                     float& t_out, float& beta_out, float& gamma_out) {
    const f3 p0 = mk3(A.x, A.y, A.z), e1 = mk3(B.x, B.y, B.z), e2 = mk3(C.x, C.y, C.z), n = mk3(A.w, B.w, C.w);
    float div = dot3(n, d);
    if (WAVE_CULL && __builtin_amdgcn_ballot_w64(!(div <= 0)) == 0ull) { t_out = 0.0f; beta_out = 0.0f; gamma_out = 0.0f; return false; }
    float idiv;
    if (FAST) idiv = rcp_refined(div);   // div <= 0 (incl. 0 -> NaN) and NaN lanes are rejected below whatever idiv is
    else idiv = rcp_plain(div);
    f3 s = sub3(o, p0);
    float beta = dot3(cross3(s, d), e2) * idiv;
    float gamma = dot3(cross3(s, e1), d) * idiv;
    float gb = gamma + beta;
    float t = dot3(cross3(s, e2), e1) * -idiv;
    // one mask per predicate, combined with & (not &&): no exec-mask branches around two-instruction tails
    int ok = !(div <= 0);
    ok &= !(beta < 0.0f) & !(beta > 1.0f);
    if (RULE == TRI_A04) ok &= !(gamma < 0.0f) & !(gamma > 1.0f) & !(gb < 0.0f) & !(gb > 1.0f);
    else ok &= !(gamma < 0.0f) & !(gb < 0.0f) & !(gb > 1.0f);
    if (RULE == TRI_A10) ok &= (t >= cmin) & (t <= cmax);
    else ok &= (t > cmin) & (t < cmax);
    t_out = t;
    beta_out = beta;
    gamma_out = gamma;
    return ok != 0;
}

// The same test for the per-lane candidate loops of the optimistic kernel (trace_cell1, LANES): the lane already knows that the sign bit
// of div is clear, and inside the guard windows (ray_guard, GridArgs::fast_ok) no operand of a comparison below can be a NaN except
// gamma + beta = (+-inf) + (-+inf), where the infinite term of the offending sign already rejects: then
//   !(beta < 0) & !(gamma < 0) & !(gb < 0)  ==  min3(beta, gamma, gb) >= 0      (v_min3_f32 drops the NaN, keeps the -inf)
//   !(beta > 1) & !(gb > 1)                 ==  max(beta, gb) <= 1              (v_max_f32 likewise keeps the +inf)
// -- five compares become two compares, one v_min3_f32 and one v_max_f32.  A lane outside the windows computes values nobody reads
// (its sample is deferred to the exact kernel).
PT_DEV bool tri_test_nn(f3 o, f3 d, float cmin, float cmax, const float4 A, const float4 B, const float4 C,
                        float& t_out, float& beta_out, float& gamma_out) {
    const f3 p0 = mk3(A.x, A.y, A.z), e1 = mk3(B.x, B.y, B.z), e2 = mk3(C.x, C.y, C.z), n = mk3(A.w, B.w, C.w);
    const float div = dot3(n, d);
    const float idiv = rcp_refined(div);
    const f3 s = sub3(o, p0);
    const float beta = dot3(cross3(s, d), e2) * idiv;
    const float gamma = dot3(cross3(s, e1), d) * idiv;
    const float gb = gamma + beta;
    const float t = dot3(cross3(s, e2), e1) * -idiv;
    const float lo = __builtin_fminf(__builtin_fminf(beta, gamma), gb);   // one v_min3_f32
    const float hi = __builtin_fmaxf(beta, gb);
    const int ok = (int)(div > 0.0f) & (int)(lo >= 0.0f) & (int)(hi <= 1.0f) & (int)(t >= cmin) & (int)(t <= cmax);
    t_out = t;
    beta_out = beta;
    gamma_out = gamma;
    return ok != 0;
}

// Groups of kTriGroup consecutive prepared triangles carry a bounding sphere {c, R'^2} behind the records (k_prepTriangles).  A ray whose
// LINE passes the centre at more than R' misses every triangle of the group geometrically, by at least a hundredth of the group's radius:
// far more than the rounding of the reference's barycentrics (relative 1e-6), so each of the reference's tests on them rejects
// (A04 / A07 code.cl interTriangle) and skipping the group changes nothing.  |v x d|^2 = |v|^2 |d|^2 - (v.d)^2 is evaluated with its
// cancellation error (<= 8 eps |v|^2 |d|^2) on the safe side; a NaN anywhere compares false = not missed.
PT_DEV bool group_missed(const Ray& ray, float dd, const float4 g) {
    const f3 v = sub3(mk3(g.x, g.y, g.z), ray.o);
    const float b = v.x * ray.d.x + v.y * ray.d.y + v.z * ray.d.z;
    const float c2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float q = __builtin_fmaf(-b, b, c2 * dd);
    return q > dd * __builtin_fmaf(1e-6f, c2, g.w);
}

// The same test, STAGED, for loops where every active lane holds a (possibly different) triangle and most lanes are rejected early
// (the frame kernels of Assign04 / 07: adjacent pixels against small triangles): after each of the reference's early-outs a wave ballot
// asks whether ANY lane is still in; if none is, the rest is skipped for the whole wave.  Lanes that are still in compute exactly the values
// the straight-line test computes (1 / div through the guarded refined reciprocal, bit-identical to the division: pt_numerics.hpp).
// `active`: lanes that hold a triangle at all.
template <int RULE>
PT_DEV bool tri_test_staged(bool active, f3 o, f3 d, float cmin, float cmax, const float4 A, const float4 B, const float4 C,
                            float& t_out, float& beta_out, float& gamma_out) {
    const f3 p0 = mk3(A.x, A.y, A.z), e1 = mk3(B.x, B.y, B.z), e2 = mk3(C.x, C.y, C.z), n = mk3(A.w, B.w, C.w);
    const float div = dot3(n, d);
    bool in = active & !(div <= 0);
    if (__builtin_amdgcn_ballot_w64(in) == 0ull) return false;
    const float idiv = rcp_exact(div, !in);
    const f3 s = sub3(o, p0);
    const float beta = dot3(cross3(s, d), e2) * idiv;
    in = in & !(beta < 0.0f) & !(beta > 1.0f);
    if (__builtin_amdgcn_ballot_w64(in) == 0ull) return false;
    const float gamma = dot3(cross3(s, e1), d) * idiv;
    const float gb = gamma + beta;
    if (RULE == TRI_A04) in = in & !(gamma < 0.0f) & !(gamma > 1.0f) & !(gb < 0.0f) & !(gb > 1.0f);
    else in = in & !(gamma < 0.0f) & !(gb < 0.0f) & !(gb > 1.0f);
    if (__builtin_amdgcn_ballot_w64(in) == 0ull) return false;
    const float t = dot3(cross3(s, e2), e1) * -idiv;
    if (RULE == TRI_A10) in = in & (t >= cmin) & (t <= cmax);
    else in = in & (t > cmin) & (t < cmax);
    t_out = t;
    beta_out = beta;
    gamma_out = gamma;
    return in;
}

// Ray-side guard of the exact cheap divisions (pt_numerics.hpp "exact division, cheaper"): |d_k| in [2^-40, 2^40] and o_k zero or
// in [2^-30, 2^20].  With the geometry-side guard (GridArgs::fast_ok, checked on the host: bounds zero or in [2^-30, 2^20],
// triangle-plane normals zero or in [2^-40, 2^40] per component) it gives
//   * every slab numerator lo-o / hi-o is zero or in [2^-53, 2^21]           -> div_exact3 == the true quotient
//   * every determinant n.d is zero or in [2^-103, 2^82] (sums of products that are zero or >= 2^-80 cancel to zero or to a
//     multiple of 2^-103)                                                     -> rcp_refined == the true reciprocal
//   * dot(d,d) is in [2^-80, 2^82]                                            -> the sphere's 1/(2a) likewise
PT_DEV bool ray_guard(const Ray& r) {
#if PT_GUARD_INT
    // the same windows on the bit patterns: for |x| the unsigned order of the bits is the order of the values, NaN and inf sort above
    // every finite value, and "zero or >= lo" is (bits - 1) >= (lo_bits - 1) in unsigned arithmetic.  Two 3-way minima, two 3-way maxima
    // and four compares instead of eighteen compares.
    auto ab = [](float x) { return __float_as_uint(x) & 0x7FFFFFFFu; };
    auto umin3 = [](uint32_t a, uint32_t b, uint32_t c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); };
    auto umax3 = [](uint32_t a, uint32_t b, uint32_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); };
    const uint32_t dx = ab(r.d.x), dy = ab(r.d.y), dz = ab(r.d.z), ox = ab(r.o.x), oy = ab(r.o.y), oz = ab(r.o.z);
    const uint32_t dlo = umin3(dx, dy, dz), dhi = umax3(dx, dy, dz);
    const uint32_t olo = umin3(ox - 1u, oy - 1u, oz - 1u), ohi = umax3(ox, oy, oz);
    return ((int)(dlo >= 0x2B800000u) & (int)(dhi <= 0x53800000u) & (int)(olo >= 0x30800000u - 1u) & (int)(ohi <= 0x49800000u)) != 0;   // 2^-40, 2^40, 2^-30, 2^20
#else
    auto dwin = [](float d) { return __builtin_fabsf(d) >= 9.094947e-13f && __builtin_fabsf(d) <= 1.0995116e12f; };          // 2^-40 .. 2^40
    auto owin = [](float o) { return o == 0.0f || (__builtin_fabsf(o) >= 9.3132257e-10f && __builtin_fabsf(o) <= 1048576.0f); };  // 0 | 2^-30 .. 2^20
    return dwin(r.d.x) && dwin(r.d.y) && dwin(r.d.z) && owin(r.o.x) && owin(r.o.y) && owin(r.o.z);
#endif
}
// SIGNED_ZERO = false: a zero numerator may come back as a zero of either sign (the two selects of div_exact3 are dropped).  Allowed
// where tmin / tmax / the exit t are only ever COMPARED (the single-cell loops: cmin <= t <= cmax); the grid walk feeds tmin into
// x = o + tmin*d and keeps the exact quotient.
template <bool SIGNED_ZERO>
PT_DEV bool slab1_fast(float lo, float hi, float o, float d, float r, BoxHit& h, float& tfar) {
    float t0, t1;
    if (SIGNED_ZERO) {
        t0 = div_exact3(lo - o, d, r);
        t1 = div_exact3(hi - o, d, r);
    } else {
        const float n0 = lo - o, n1 = hi - o;
#if PT_PLAIN_DIV
        t0 = n0 / d;
        t1 = n1 / d;
#else
        const float q0 = n0 * r, q1 = n1 * r;
        t0 = __builtin_fmaf(__builtin_fmaf(-d, q0, n0), r, q0);
        t1 = __builtin_fmaf(__builtin_fmaf(-d, q1, n1), r, q1);
#endif
        // Inside the guard windows both quotients are finite (no NaN), and correctly rounded division is monotonic, so "the plane
        // the ray meets first" (d < 0 ? t1 : t0) IS min(t0, t1) and the other one max(t0, t1); OpenCL's min / max select forms then
        // agree with v_min_f32 / v_max_f32 up to the sign of a zero, which nothing downstream can see (compare-only).  Six full-rate
        // instructions instead of three compares and six selects per slab pair... per box: 12 instead of 21 half-rate ones.
        const float tn = __builtin_fminf(t0, t1), tf = __builtin_fmaxf(t0, t1);
        tfar = tf;
        h.tmin = __builtin_fmaxf(tn, h.tmin);
        h.tmax = __builtin_fminf(tf, h.tmax);
        return !(h.tmin > h.tmax);
    }
    const bool neg = d < 0;
    float tn = neg ? t1 : t0;
    float tf = neg ? t0 : t1;
    tfar = tf;
    h.tmin = cl_max(tn, h.tmin);
    h.tmax = cl_min(tf, h.tmax);
    return !(h.tmin > h.tmax);
}
// inter_aabb (pt_device.hpp); FAST: the six slab quotients share three refined reciprocals, and the boxes of all the sets a ray is tested
// against share those: RayRcp is made once per ray (closest_all / direct_all), not once per set
struct RayRcp { float x, y, z; };
template <bool FAST>
PT_DEV RayRcp ray_rcp(const Ray& r) {
    RayRcp q;
    q.x = q.y = q.z = 0.0f;
    if (FAST) { q.x = rcp_refined(r.d.x); q.y = rcp_refined(r.d.y); q.z = rcp_refined(r.d.z); }
    return q;
}
template <bool FAST, bool SIGNED_ZERO = true>
PT_DEV BoxHit inter_aabb_t(const Ray& r, const RayRcp& q, const Box& b) {
    if (FAST && !SIGNED_ZERO) {
        // compare-only users (the single-cell sets).  tmin only grows and tmax only shrinks from slab to slab and neither is ever a NaN
        // (min / max drop a NaN operand and the chains start at 0 and inf), so the reference's three early returns (code.cl:351-354,
        // 366-369, 381-384) say no exactly when the final pair does: one comparison, and the chains as v_max3_f32 / v_min3_f32.  Inside
        // the guard windows every quotient is finite, so the chain's leading min(inf, .) is the identity; the sign of a zero is invisible
        // to comparisons (slab1_fast).
        BoxHit h;
#if PT_PLAIN_DIV
        const float x0 = (b.lo.x - r.o.x) / r.d.x, x1 = (b.hi.x - r.o.x) / r.d.x, y0 = (b.lo.y - r.o.y) / r.d.y, y1 = (b.hi.y - r.o.y) / r.d.y;
        const float z0 = (b.lo.z - r.o.z) / r.d.z, z1 = (b.hi.z - r.o.z) / r.d.z;
#else
        auto quo = [](float num, float d, float rr) { const float q0 = num * rr; return __builtin_fmaf(__builtin_fmaf(-d, q0, num), rr, q0); };
        const float x0 = quo(b.lo.x - r.o.x, r.d.x, q.x), x1 = quo(b.hi.x - r.o.x, r.d.x, q.x);
        const float y0 = quo(b.lo.y - r.o.y, r.d.y, q.y), y1 = quo(b.hi.y - r.o.y, r.d.y, q.y);
        const float z0 = quo(b.lo.z - r.o.z, r.d.z, q.z), z1 = quo(b.hi.z - r.o.z, r.d.z, q.z);
#endif
        h.tfx = __builtin_fmaxf(x0, x1); h.tfy = __builtin_fmaxf(y0, y1); h.tfz = __builtin_fmaxf(z0, z1);
        h.tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fmaxf(__builtin_fminf(z0, z1), 0.0f));
        h.tmax = __builtin_fminf(__builtin_fminf(h.tfx, h.tfy), h.tfz);
        h.v = !(h.tmin > h.tmax);
        return h;
    }
    if (FAST) {
        BoxHit h;
        h.tmin = 0.0f;
        h.tmax = PT_INF;
        const bool okx = slab1_fast<SIGNED_ZERO>(b.lo.x, b.hi.x, r.o.x, r.d.x, q.x, h, h.tfx);
        const bool oky = slab1_fast<SIGNED_ZERO>(b.lo.y, b.hi.y, r.o.y, r.d.y, q.y, h, h.tfy);
        const bool okz = slab1_fast<SIGNED_ZERO>(b.lo.z, b.hi.z, r.o.z, r.d.z, q.z, h, h.tfz);
        h.v = okx && oky && okz;
        return h;
    }
    return inter_aabb(r, b);
}
template <bool FAST, bool SIGNED_ZERO = true>
PT_DEV BoxHit inter_aabb_t(const Ray& r, const Box& b) { return inter_aabb_t<FAST, SIGNED_ZERO>(r, ray_rcp<FAST>(r), b); }

struct SphereRay { float a, inv2a; };  // ray-only part of the quadratic (A10 code.cl:203, 218)
template <bool FAST>
PT_DEV SphereRay sphere_ray(f3 d) {
    SphereRay r;
    r.a = dot3(d, d);
    r.inv2a = FAST ? rcp_refined(2.0f * r.a) : rcp_plain(2.0f * r.a);
    return r;
}
// ORDERED (the optimistic kernel): with a = d.d > 0 the two roots come out ordered, t0 = (-b - sq) / 2a <= t1 = (-b + sq) / 2a (sq >= 0 or NaN; sums
// and products by a positive factor are monotonic under rounding), so fmin / fmax are the identity; the exact kernel keeps them (a ray
// with infinite coordinates can make exactly one root a NaN).  `dis < 0` needs no test of its own anywhere: its square root is a NaN
// (sqrt_core's v_sqrt_f32 and both neighbour residuals; the rare-lane path for tiny magnitudes is the library's sqrt), both roots are
// NaNs and every window comparison below is false -- the reference's early return (code.cl:206-209) taken by arithmetic.
// WAVE_SKIP (the wave-uniform loops: every lane holds the SAME sphere): when the discriminant is negative in every lane the square root and
// the window tests are skipped for the wave -- the reference's early return (code.cl:206-209) taken when the whole wave takes it; a small
// sphere is missed by all 64 rays of an incoherent wave about one time in five.
template <bool ORDERED = false, bool WAVE_SKIP = false, int COUNTER = -1>
PT_DEV bool sph_test(f3 o, f3 d, const SphereRay& sr, float cmin, float cmax, const float4 sph, float& t_out) {
    f3 omc = sub3(o, ld3(sph));
    float b = 2.0f * dot3(omc, d);
    float c = dot3(omc, omc) - sph.w;
    float dis = cl_mad(-4.0f * c, sr.a, b * b);
    if (WAVE_SKIP && __builtin_amdgcn_ballot_w64(!(dis < 0.0f)) == 0ull) { t_out = 0.0f; return false; }
    if (COUNTER >= 0) pt_count(COUNTER);
    float sq = cl_sqrt(dis);
    float t0 = (-b - sq) * sr.inv2a;
    float t1 = (-b + sq) * sr.inv2a;
    float tmin = ORDERED ? t0 : cl_fmin(t0, t1);
    float tmax = ORDERED ? t1 : cl_fmax(t0, t1);
    const int in0 = (tmin >= cmin) & (tmin <= cmax);
    const int in1 = (tmax >= cmin) & (tmax <= cmax);
    t_out = in0 ? tmin : tmax;
    return (in0 | in1) != 0;
}

// axis_setup (pt_device.hpp) with the optimistic kernel's divisions: every quotient is div_exact3 on a refined reciprocal.  The
// denominators are the ray direction (ray_guard), the slab count and the slab width; numerators and the slab width are checked
// here, per lane, and a lane outside the windows marks its sample for the exact kernel.
PT_DEV bool num_window(float v) { const float a = __builtin_fabsf(v); return ((int)(v == 0.0f) | ((int)(a >= 8.6736174e-19f) & (int)(a <= 1.1529215e18f))) != 0; }   // 0 | 2^-60 .. 2^60
PT_DEV bool den_window(float v) { const float a = __builtin_fabsf(v); return a >= 9.094947e-13f && a <= 1.0995116e12f; }                    // 2^-40 .. 2^40
// The slab width delta = (hi - lo) / n and 1 / delta depend on the set alone: the host computes them once (GridArgs::delta / rdelta,
// correctly rounded, which is what div_exact3 / rcp_refined give inside their windows; GridArgs::walk_ok says the windows hold).
template <bool FAST>
PT_DEV Axis axis_setup_t(float o, float d, float tmin, float lo, float hi, uint32_t n, float delta, float rdelta, float rd, bool& defer) {   // rd = rcp_refined(d): the ray's, made once (RayRcp)
    if (!FAST) return axis_setup(o, d, tmin, lo, hi, n);
#if PT_PLAIN_DIV
    // the default contract (libmirt_default.so): the slab width is the reference's own quotient (code.cl:699) in THIS contract's 2.5-ulp division; the host's
    // GridArgs::delta is the correctly rounded one, which is not always the same float.  (The argument is wave-uniform: the compiler divides once per walk.)
    delta = (hi - lo) / (float)n;
#endif
    Axis a;
    const float x = cl_fma(tmin, d, o);                    // code.cl:698
    const float num0 = x - lo;
    {   // if (slab < 0) slab = 0; if (slab >= n) slab = n - 1 (code.cl:699-700), spelled so that it is one v_med3_i32
        const int q = f2i(div_exact3_anyzero(num0, delta, rdelta)), top = (int)(n - 1u), q0 = q > 0 ? q : 0;   // (a zero of either sign converts to 0)
        a.slab = q0 < top ? q0 : top;
    }
    const bool fwd = d >= 0;
    a.dslab = fwd ? 1 : -1;
    a.limit = fwd ? (int)n : -1;
    a.dt = div_exact3_anyzero(delta, cl_fabs(d), cl_fabs(rd));   // rcp_refined is odd in d: every step is sign-symmetric under RNE; delta != 0 (GridArgs::walk_ok)
    const float xnext = cl_fma((float)(a.slab + (fwd ? 1 : 0)), delta, lo);   // code.cl:706
    const float num1 = xnext - o;
    // (the sign of a zero tnext is invisible: the walk compares it, adds the positive dt to it and passes it through v_min3_f32 into windows that are only compared)
    a.tnext = div_exact3_anyzero(num1, d, rd);
    // (bitwise on purpose: with || and && the compiler branches around the second window of a lane that already defers -- a dozen scalar instructions
    // and a block boundary per axis of every walk, to save six compares nobody waits for)
    defer = ((int)defer | (int)!((int)num_window(num0) & (int)num_window(num1))) != 0;
    return a;
}

// all ones where the sign bit of w is set, else zero.  As the instruction itself: from the C shift the compiler makes a compare, a move of the
// mask into a VGPR, a select and an or (four instructions and an s_nop where the sweep needs two: this and one v_bitop3_b32)
PT_DEV uint32_t sign_word(float w) {
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(w));
    return r;
}

// The block's dynamic LDS (launch_fused sizes it): [cooperative-walk exchange area (pt_trace_coop.hpp)] [staged cell-offset tables]
// [staged single-cell triangle sets]: GridArgs::lds_off indexes it.  The walks read it through THIS symbol, so the compiler knows the
// address space and emits ds_read (through a generic pointer selected at run time it emitted flat_load pairs).
extern __shared__ __attribute__((aligned(16))) uint32_t pt_lds_dyn[];   // read through float4 casts (ds_read_b128): every sub-area starts on a multiple of four words

PT_DEV Box set_box_of(const GridArgs& S) {
    Box b;
    b.lo = mk3(S.bound[0], S.bound[1], S.bound[2]);
    b.hi = mk3(S.bound[4], S.bound[5], S.bound[6]);
    return b;
}
// The t at which a ray leaves the single cell of an n == 1 set: axis_setup with n == 1 -- slab = 0, the cell exit is the far face as the
// reference computes it, lo + (0 + (d>=0)) * ((hi-lo)/1)   (A10 code.cl:699-707)
PT_DEV float cell1_exit(const Ray& ray, const BoxHit& bh, const GridArgs& S) {
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
    return cl_min(cl_min(tn[0], tn[1]), tn[2]);   // one v_min3_f32
}

// One primitive set.  KIND / ANY as in pt_device.hpp.
// trace_cell1: n == 1, a single cell, every lane walks the same list -> wave-uniform loop, scalar loads.
// FLAG_ONLY (with ANY): the caller only asks whether the ray is blocked (the fused pass: sceneRender compares mint with maxt and nothing
// else of the shadow ray survives the kernel).  The loop then keeps one flag instead of the hit record: idx != UINT32_MAX says blocked, t
// is not delivered.
// LANES (optimistic kernel, triangles, a set whose prepared records the block staged in LDS: S.lds_off): per-lane candidate lists.
// The wave-uniform loop evaluates every test for every lane although the reference leaves half of them after five operations
// (div <= 0: the triangle faces away, code.cl:256-260) -- on an incoherent wave some lane always needs the triangle, so the whole
// wave pays for it.  Here a first wave-uniform sweep computes div = dot(n, d) per triangle (n by scalar loads from the compact
// array k_prepTriangles leaves behind the records) and shifts a sign bit into a per-lane word; then every lane walks ITS OWN
// candidates -- sign bit clear -- in list order, fetching each record from LDS by ds_read_b128 (twelve-dword records: up to sixteen
// distinct triangles are conflict-free).  In a closed room 5-7 of cornell.xml's 12 triangles survive the sweep.  Same predicates, same
// arithmetic, same order within a lane (ties on t go to the lower index exactly as in the reference's loop).
template <int KIND, bool ANY, int RULE = TRI_A10, bool FAST = false, bool FLAG_ONLY = false, bool LANES = false>
PT_DEV Hit trace_cell1(const Ray& ray, const BoxHit& bh, const GridArgs& S) {
    Hit ch;
    ch.idx = UINT32_MAX;
    ch.t = ray.maxt;
    ch.beta = 0.0f;
    ch.gamma = 0.0f;
    SphereRay sr;
    if (KIND == SPHERES) sr = sphere_ray<FAST>(ray.d);
    const float cmin = bh.tmin;
    const float cmax = cell1_exit(ray, bh, S);
    const uint32_t begin = __builtin_amdgcn_readfirstlane(ldc_u32(S.off, 0));
    const uint32_t end = __builtin_amdgcn_readfirstlane(ldc_u32(S.off, 1));
    bool done = false;
    if (LANES && KIND == TRIANGLES && FAST && RULE == TRI_A10 && S.lds_off != kNoLds && begin == 0u) {
        // the sweep's plane list (k_planeList; layout in pt_launch.hpp): a 64-byte header {groups[4], first[4], Gmax[4], Hmax[4]} (one slot per chunk of 32
        // records), then 64-byte groups of entries -- one s_load_dwordx16 each: four axis planes {q, n_a, mask, back mask} or two general ones
        // {n.xyz, k = p0 . n, mask, back mask, 0, 0}; mask = the chunk's records in that plane, record c0 + j at bit 31 - j
        const pt_v16f PT_CONST_AS* pn = (const pt_v16f PT_CONST_AS*)((const char PT_CONST_AS*)S.pnorm + 64);
        const uint32_t lds_bytes = S.lds_off * 4u;
        // PLANE WINDOW (the second filter of the sweep; the first is the facing test div > 0).  A hit needs cmin <= t <= cmax and t < maxt
        // (code.cl:273-280 and the callers' comparisons), where the reference's t = fl(dot(cross(s, e2), e1) * -(1 / div)).  In exact arithmetic
        // that numerator is s . (e2 x e1) = o . n - p0 . n, which the sweep gets in three fused operations: sn = fma(n.z, o.z, fma(n.y, o.y,
        // fma(n.x, o.x, -k))) -- or, for a plane {x_a = q} (an AXIS entry: n = n_a e_a exactly, every vertex at x_a = q), sn = n_a (o_a - q) and
        // div = n_a d_a, which is what the reference's three-term dot product comes to when two terms are products with zero (the sign of a
        // zero result aside, and a zero div is a reject either way).  Both evaluations of the numerator stay within 14 u E (|o|_1 + |p0|_1) of
        // each other (u = 2^-24, E = |e1|_1 |e2|_1: the rounding of s, of the two cross products, of the dot products, of n and of k, term by
        // term; the axis form has two roundings, each relative to |n_a| (|o_a| + |q|) <= E (|o|_1 + |p0|_1)); the sweep allows
        // M = 2^-17 E (|o|_1 + |p0|_1), nine times that, never less than 2^-56, and widens the window by 2^-20 relative, which also covers the
        // three roundings between the reference's numerator and its t and those of the two tests below (every plane of a chunk gets the
        // chunk's LARGEST G and H -- a wider margin keeps more, never less).  With div > 0:
        //   -sn - M > hi+ * div,  hi+ = max(min(cmax, maxt) * (1 + 2^-20), 2^-100)   =>   the reference's t > min(cmax, maxt): no hit
        //   -sn + M < lo- * div,  lo- = cmin * (1 - 2^-20)                            =>   the reference's t < cmin, and at least
        //        2^-57 / 2^82 in magnitude when cmin = 0 (no underflow to a -0 that would pass t >= 0): no hit
        // so such a triangle is not a candidate: every test the lane skips is one the reference's own window rejects, and the tests it
        // does run are the reference's.  w = min3(div, c, -a) carries the verdict in its sign (min3 drops a NaN operand: then the triangle
        // stays a candidate; lanes outside the guard windows are deferred to the exact kernel whatever they compute here).  The BACK mask of
        // an entry holds the records with the reversed normal: -n, -k negate div and sn exactly, so their verdict is
        // min3(-div, -fma(hi+, div, sn - M), fma(lo-, div, sn + M)) -- two more fused operations on sums the front verdict already made.
        // In cornell.xml a closest query keeps 2-4 of the 5-7 front-facing triangles, a shadow query 0-2.
        const float o1 = __builtin_fabsf(ray.o.x) + __builtin_fabsf(ray.o.y) + __builtin_fabsf(ray.o.z);
        const float hi_p = __builtin_fmaxf(__builtin_fminf(cmax, ray.maxt) * 1.00000095367431640625f, 0x1p-100f);
        const float lo_m = cmin * 0.99999904632568359375f;
        pt_count(ANY ? PC_Q_SHADOW : PC_Q_CLOSEST); pt_count(ANY ? PC_Q_SHADOW_LANES : PC_Q_CLOSEST_LANES, true);
        for (uint32_t c0 = 0u; c0 < end; c0 += 32u) {
            const uint32_t groups = ldc_u32(S.pnorm, c0 >> 5);
            const pt_v16f PT_CONST_AS* g = pn + ldc_u32(S.pnorm, 4u + (c0 >> 5));
            const float Mu = cl_fma(__uint_as_float(ldc_u32(S.pnorm, 8u + (c0 >> 5))), o1, __uint_as_float(ldc_u32(S.pnorm, 12u + (c0 >> 5))));
            uint32_t cand = 0u;
            // the verdict of one plane from its div and sn: candidates |= mask where w >= 0 (sign bit clear)
            auto verdict = [&](float div, float sn, uint32_t mask, uint32_t back) {
                const float sp = sn + Mu, sm = sn - Mu;
                const float w = __builtin_fminf(__builtin_fminf(div, cl_fma(hi_p, div, sp)), -cl_fma(lo_m, div, sm));   // one v_min3_f32
                cand |= mask & ~sign_word(w);
                if (back != 0u) {   // wave-uniform (an SGPR)
                    asm volatile("" ::: "memory");
                    const float wb = __builtin_fminf(__builtin_fminf(-div, -cl_fma(hi_p, div, sm)), cl_fma(lo_m, div, sp));
                    cand |= back & ~sign_word(wb);
                }
            };
#define PT_AXIS_GROUPS(SHIFT, OA, DA)                                                                                   \
            for (uint32_t i = 0; i < ((groups >> SHIFT) & 255u); ++i, ++g) {                                            \
                const pt_v16f v = *g;                                                                                   \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                         \
                    const uint32_t mask = __float_as_uint(v[4 * k + 2]), back = __float_as_uint(v[4 * k + 3]);          \
                    if ((mask | back) == 0u) continue;   /* a padding entry: skipped on a scalar branch */              \
                    asm volatile("" ::: "memory");                                                                      \
                    pt_count(ANY ? PC_SWEEP_SHADOW : PC_SWEEP_CLOSEST);                                                 \
                    verdict(v[4 * k + 1] * (DA), v[4 * k + 1] * ((OA) - v[4 * k]), mask, back);                         \
                }                                                                                                       \
            }
            PT_AXIS_GROUPS(0, ray.o.x, ray.d.x)
            PT_AXIS_GROUPS(8, ray.o.y, ray.d.y)
            PT_AXIS_GROUPS(16, ray.o.z, ray.d.z)
#undef PT_AXIS_GROUPS
            for (uint32_t i = 0; i < groups >> 24; ++i, ++g) {
                const pt_v16f v = *g;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint32_t mask = __float_as_uint(v[8 * k + 4]), back = __float_as_uint(v[8 * k + 5]);
                    if ((mask | back) == 0u) continue;
                    asm volatile("" ::: "memory");   // keeps the skip a scalar branch (the compiler would rather compute the padding entry and select)
                    pt_count(ANY ? PC_SWEEP_SHADOW : PC_SWEEP_CLOSEST);
                    const float div = dot3(mk3(v[8 * k], v[8 * k + 1], v[8 * k + 2]), ray.d);
                    const float sn = cl_fma(v[8 * k + 2], ray.o.z, cl_fma(v[8 * k + 1], ray.o.y, cl_fma(v[8 * k], ray.o.x, -v[8 * k + 3])));
                    verdict(div, sn, mask, back);
                }
            }
            if (ANY && done) cand = 0u;            // (a set of more than 32 triangles: a blocked lane sits the later sweeps out)
            while (cand != 0u) {
                pt_count(ANY ? PC_TRIPS_SHADOW : PC_TRIPS_CLOSEST); pt_count(ANY ? PC_TRIP_LANES_SHADOW : PC_TRIP_LANES_CLOSEST, true);
                const uint32_t k = (uint32_t)__builtin_clz(cand);
                cand &= ~(0x80000000u >> k);
                const uint32_t i = c0 + k;
                const float4* __restrict__ q = (const float4*)((const char*)pt_lds_dyn + (lds_bytes + __umul24(i, 48u)));
                float ti, b, gm;
                const bool hit = tri_test_nn(ray.o, ray.d, cmin, cmax, q[0], q[1], q[2], ti, b, gm);
                if (ANY && FLAG_ONLY) {
                    if (hit && ti < ray.maxt) { done = true; cand = 0u; }
                    continue;
                }
                const bool better = (int)hit & (int)(ti < ch.t);
                ch.t = better ? ti : ch.t;
                ch.idx = better ? i : ch.idx;
                ch.beta = better ? b : ch.beta;
                ch.gamma = better ? gm : ch.gamma;
                if (ANY && better) { done = true; cand = 0u; }
            }
            if (ANY && __builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        }
        if (ANY && FLAG_ONLY && done) ch.idx = 0u;
        return ch;
    }
    if (KIND == SPHERES) { pt_count(ANY ? PC_SPH_Q_SHADOW : PC_SPH_Q_CLOSEST); pt_count(ANY ? PC_SPH_Q_SHADOW_LANES : PC_SPH_Q_CLOSEST_LANES, true); }
    const pt_v4f PT_CONST_AS* p = (const pt_v4f PT_CONST_AS*)S.prims + (size_t)begin * (KIND == SPHERES ? 1u : 3u);   // one running pointer: immediate-offset scalar loads
#if PT_UNROLL_UNIFORM > 1
#pragma unroll PT_UNROLL_UNIFORM
#endif
    for (uint32_t i = begin; i < end; ++i, p += (KIND == SPHERES ? 1 : 3)) {
        float ti, b = 0.0f, gm = 0.0f;
        bool hit;
        if (KIND == SPHERES) {
            pt_count(ANY ? PC_SPH_TESTS_SHADOW : PC_SPH_TESTS_CLOSEST);
            hit = sph_test<FAST, PT_SPHERE_WAVE_SKIP != 0, ANY ? PC_SPH_ROOTS_SHADOW : PC_SPH_ROOTS_CLOSEST>(ray.o, ray.d, sr, cmin, cmax, ldc4((const void*)p, 0), ti);
        } else {
            hit = tri_test<RULE, FAST, PT_UNIFORM_CULL != 0>(ray.o, ray.d, cmin, cmax, ldc4((const void*)p, 0), ldc4((const void*)p, 1), ldc4((const void*)p, 2), ti, b, gm);
        }
        if (ANY && FLAG_ONLY) {
            done = done | ((int)hit & (int)(ti < ray.maxt));   // the first hit ends the reference's loop: ch.t is still maxt when it is compared
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            continue;
        }
        const bool better = (int)!done & (int)hit & (int)(ti < ch.t);
        ch.t = better ? ti : ch.t;          // selects, not a branch: some lane of an incoherent wave almost always hits
        ch.idx = better ? i : ch.idx;
        ch.beta = better ? b : ch.beta;
        ch.gamma = better ? gm : ch.gamma;
        if (ANY) {
            done = done || better;
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;  // every lane of the wave is blocked
        }
    }
    if (ANY && FLAG_ONLY && done) ch.idx = 0u;
    return ch;
}

// The fused pass stages the cell-offset tables of its grid sets in the block's dynamic LDS.  LDS_TABLES is a template parameter because
// a run-time choice of address space degenerates into flat loads: k_fusedPass<*, 1> has every table staged, k_fusedPass<*, 2> (a scene
// whose tables do not fit) reads them all from memory.
// cell -> [begin, end) of a grid set: one 8-byte LDS read when the table is staged, else two dwords from memory
template <bool LDS_TABLES>
PT_DEV void cell_range(const GridArgs& S, const uint32_t* __restrict__ off, uint32_t cell, uint32_t& i, uint32_t& end) {
    if (LDS_TABLES) {   // compile-time: a run-time choice between the two gets merged back into flat loads
        const uint32_t k = S.lds_off + cell;
        i = pt_lds_dyn[k];
        end = pt_lds_dyn[k + 1u];
    } else {
        i = off[cell];
        end = off[cell + 1];
    }
}

// trace_dda: n > 1, per-lane 3-axis DDA.
template <int KIND, bool ANY, int RULE = TRI_A10, bool FAST = false, bool LDS_TABLES = false>
PT_DEV Hit trace_dda(const Ray& ray, const BoxHit& bh, const GridArgs& S, bool& defer) {
    const float4* __restrict__ prims = (const float4*)S.prims;
    const uint32_t* __restrict__ off = (const uint32_t*)S.off;
    Hit ch;
    ch.idx = UINT32_MAX;
    ch.t = ray.maxt;
    ch.beta = 0.0f;
    ch.gamma = 0.0f;
    SphereRay sr;
    if (KIND == SPHERES) sr = sphere_ray<FAST>(ray.d);
    // Per-axis walk state kept to what the loop reads: the next plane's t, the step in t, the slab index.  The step direction and the
    // index at which the ray leaves the grid follow from the sign of d, which is still in a register (code.cl:701-705: d >= 0 ? +1, n : -1, -1).
    float tnx, tny, tnz, dtx, dty, dtz;
    int sx, sy, sz;
    {
        if (FAST) defer = defer || S.walk_ok == 0u;
        const Axis ax = axis_setup_t<FAST>(ray.o.x, ray.d.x, bh.tmin, S.bound[0], S.bound[4], S.n, S.delta[0], S.rdelta[0], FAST ? rcp_refined(ray.d.x) : 0.0f, defer);
        const Axis ay = axis_setup_t<FAST>(ray.o.y, ray.d.y, bh.tmin, S.bound[1], S.bound[5], S.n, S.delta[1], S.rdelta[1], FAST ? rcp_refined(ray.d.y) : 0.0f, defer);
        const Axis az = axis_setup_t<FAST>(ray.o.z, ray.d.z, bh.tmin, S.bound[2], S.bound[6], S.n, S.delta[2], S.rdelta[2], FAST ? rcp_refined(ray.d.z) : 0.0f, defer);
        tnx = ax.tnext; tny = ay.tnext; tnz = az.tnext;
        dtx = ax.dt; dty = ay.dt; dtz = az.dt;
        sx = ax.slab; sy = ay.slab; sz = az.slab;
    }
    const int nn = (int)S.n;
    // The reference walks cell by cell and, inside a cell, primitive by primitive (two nested loops per work-item).  Run that way
    // on a 64-lane wave every outer iteration costs the LONGEST list any lane holds.  Same per-lane sequence, re-phased:
    //   phase A: every lane whose list is exhausted closes its cell (stop on a hit, else step) and opens the next one -- cheap
    //            iterations, repeated until each live lane holds a primitive or has left the grid;
    //   phase B: every live lane tests ONE primitive.
    // A lane's tests, their order and their [cmin, cmax] windows are exactly those of the nested loops.
    const uint32_t zs = S.n * S.n, ys = S.n;   // n <= 1024 (check_grid): 24-bit multiplies are exact
    float t = bh.tmin;
    float cmin = t;
    // the three exits through v_min_f32 (OpenCL min() IS that instruction under the contract, pt_numerics.hpp)
    float cmax = cl_min(cl_min(tnx, tny), tnz);
    uint32_t cell = __umul24((uint32_t)sz, zs) + __umul24((uint32_t)sy, ys) + (uint32_t)sx;
    uint32_t i, end;
    cell_range<LDS_TABLES>(S, off, cell, i, end);
    for (;;) {
        bool alive = true;
        while (i == end) {
            // close the cell: a hit inside it ends the walk (code.cl:768-771); else step the axis whose plane was reached.  The walk
            // also ends where the next cell would start at or beyond the ray's end (t >= maxt): a hit needs cmin <= t < maxt and cmin only
            // grows, so nothing the reference computes past that point survives.
            // Kept as the reference's if / else-if / else chain: the walk is VALU-bound at a quarter of the lanes, and the
            // branches cost scalar instructions, which are not the bottleneck (selects instead: 72.2 -> 77.6 ms on cornell_teapot3)
            if (ch.idx != UINT32_MAX) { alive = false; break; }
            t = cmax;
            if (t == tnx) {
                tnx += dtx;
                const bool fwd = ray.d.x >= 0;
                sx += fwd ? 1 : -1;
                if (t >= bh.tmax || t >= ray.maxt || sx == (fwd ? nn : -1)) { alive = false; break; }
            } else if (t == tny) {
                tny += dty;
                const bool fwd = ray.d.y >= 0;
                sy += fwd ? 1 : -1;
                if (t >= bh.tmax || t >= ray.maxt || sy == (fwd ? nn : -1)) { alive = false; break; }
            } else {
                tnz += dtz;
                const bool fwd = ray.d.z >= 0;
                sz += fwd ? 1 : -1;
                if (t >= bh.tmax || t >= ray.maxt || sz == (fwd ? nn : -1)) { alive = false; break; }
            }
            cmin = t;
            cmax = cl_min(cl_min(tnx, tny), tnz);
            cell = __umul24((uint32_t)sz, zs) + __umul24((uint32_t)sy, ys) + (uint32_t)sx;
            cell_range<LDS_TABLES>(S, off, cell, i, end);
        }
        if (!alive) break;
        float ti, b = 0.0f, gm = 0.0f;
        bool hit;
        if (KIND == SPHERES) {
            hit = sph_test(ray.o, ray.d, sr, cmin, cmax, prims[i], ti);
        } else {
            const float4* __restrict__ p = prims + 3u * (size_t)i;
            hit = tri_test<RULE, FAST>(ray.o, ray.d, cmin, cmax, p[0], p[1], p[2], ti, b, gm);
        }
        const bool better = (int)hit & (int)(ti < ch.t);
        ch.t = better ? ti : ch.t;
        ch.idx = better ? i : ch.idx;
        ch.beta = better ? b : ch.beta;
        ch.gamma = better ? gm : ch.gamma;
        ++i;
        if (ANY && better) break;
    }
    return ch;
}

template <int KIND, bool ANY, int RULE = TRI_A10, bool FAST = false>
PT_DEV Hit trace_set(const Ray& ray, const BoxHit& bh, const GridArgs& S, bool& defer) {
    if (S.n == 1u) return trace_cell1<KIND, ANY, RULE, FAST>(ray, bh, S);
    return trace_dda<KIND, ANY, RULE, FAST>(ray, bh, S, defer);
}

}  // namespace pt
