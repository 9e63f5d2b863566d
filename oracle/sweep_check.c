/* oracle/sweep_check.c -- TEST INFRASTRUCTURE (part of liboracle.so; the product never links it).
 *
 * A CPU restatement of the candidate sweep's PLANE WINDOW (2015-raytracing_amd/csrc/pt_trace.hpp trace_cell1, LANES; the plane list of
 * k_planeList in pt_kernels_fused.hip: general planes, AXIS planes, and the BACK verdict of a plane listed with the reversed normal)
 * beside the reference's interTriangle (A10 code.cl:250-288) under the numerics contract of
 * cl_numerics.h, to check the one claim the product's speed-up rests on:
 *
 *      the sweep drops a triangle  ==>  the reference's own test rejects it  (for the window the caller compares t with)
 *
 * oracle_sweep_check() draws rays, triangles and windows -- random ones and ADVERSARIAL ones: a ray aimed at a point of the triangle,
 * the window edges cmin / cmax / maxt placed on the reference's own t for that pair and a few ulps either side, geometry scaled over
 * forty octaves, needle triangles, near-parallel rays -- and counts the violations (dropped by the sweep, accepted by the reference).
 * tests/test_sweep_filter.py asserts the count is zero and that the sweep still drops most of what the reference rejects (it is a
 * filter, not a constant `true`).  The margin's derivation is in pt_trace.hpp; this is its empirical side, runnable without a GPU.
 * The margin checked is each triangle's OWN (G, H); the kernel gives every plane of a 32-record chunk the chunk's largest pair,
 * which keeps a superset of what is kept here: the check is on the tightest margin the product can use.
 *
 * Second half (oracle_plane_list, oracle_sweep_words): the LAYOUT the kernel consumes -- k_planeList restated word for word (classes, merged
 * masks, back masks, 64-byte groups, the chunk's largest margins) and the sweep's walk over it, so tests/test_sweep_filter.py can check on the
 * CPU that candidate bit 31 - j of chunk c is record 32 c + j and that every record sits in exactly one mask, and tests/test_gpu_parity.py that
 * the device builds the same bytes (mirt_debug_prepared). */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "cl_numerics.h"

typedef struct { float x, y, z; } v3;

static inline v3 v3sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline float v3dot(v3 a, v3 b) { return cln_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
static inline v3 v3cross(v3 a, v3 b) {
    float A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z}, R[3];
    cln_cross3(A, B, R);
    v3 r = {R[0], R[1], R[2]};
    return r;
}
static inline float l1(v3 a) { return (fabsf(a.x) + fabsf(a.y)) + fabsf(a.z); }

/* the reference's test on (p0, e1 = p1 - p0, e2 = p2 - p0), A10 rule: closed t interval; returns accept and t */
static int ref_test(v3 o, v3 d, float cmin, float cmax, float maxt, v3 p0, v3 e1, v3 e2, float* t_out) {
    const v3 n = v3cross(e2, e1);
    const float div = v3dot(n, d);
    *t_out = NAN;
    if (div <= 0.0f) return 0;
    const float idiv = 1.0f / div;
    const v3 s = v3sub(o, p0);
    const float beta = v3dot(v3cross(s, d), e2) * idiv;
    if (beta < 0.0f || beta > 1.0f) return 0;
    const float gamma = v3dot(v3cross(s, e1), d) * idiv;
    const float gb = gamma + beta;
    if (gamma < 0.0f || gb < 0.0f || gb > 1.0f) return 0;
    const float t = v3dot(v3cross(s, e2), e1) * -idiv;
    *t_out = t;
    return t >= cmin && t <= cmax && t < maxt;
}
/* the same without the window: the t the reference computes whenever div > 0 */
static float ref_t(v3 o, v3 d, v3 p0, v3 e1, v3 e2) {
    const v3 n = v3cross(e2, e1);
    const float div = v3dot(n, d);
    if (!(div > 0.0f)) return NAN;
    const v3 s = v3sub(o, p0);
    return v3dot(v3cross(s, e2), e1) * -(1.0f / div);
}

/* k_planeList: the margin constants of one record */
static void plane_margin(v3 p0, v3 e1, v3 e2, float* G, float* H) {
    const float up = 1.0000002384185791015625f;
    const float E = ((l1(e1)) * up) * ((l1(e2)) * up) * up;
    *G = 0x1p-17f * E;
    *H = fmaxf(*G * (l1(p0) * up) * up, 0x1p-56f);
}
/* k_planeList: the class of a record -- 0 / 1 / 2: an axis plane {x_a = p0_a} (n has two zero components, both edges exactly zero along a);
 * 3: general */
static int plane_class(v3 p0, v3 e1, v3 e2, v3 n) {
    const float nn[3] = {n.x, n.y, n.z}, a1[3] = {e1.x, e1.y, e1.z}, a2[3] = {e2.x, e2.y, e2.z};
    int cl = 3;
    (void)p0;
    for (int a = 0; a < 3; ++a)
        if (nn[a] != 0.0f && nn[(a + 1) % 3] == 0.0f && nn[(a + 2) % 3] == 0.0f && a1[a] == 0.0f && a2[a] == 0.0f) cl = a;
    return cl;
}
static inline float canon0(float v) { return v == 0.0f ? 0.0f : v; }   /* -0 -> +0, as the list stores a normal's zero components */
/* trace_cell1's verdicts from a plane's div and sn: 1 = candidate (sign bit of w clear).  front: the records of the entry's mask; back: those
 * of its back mask (the same plane listed with the reversed normal) */
static int verdict_front(float div, float sn, float M, float hi_p, float lo_m) {
    const float sp = sn + M, sm = sn - M;
    const float w = cln_min(cln_min(div, fmaf(hi_p, div, sp)), -fmaf(lo_m, div, sm));
    return (cln_bits(w) >> 31) == 0u;
}
static int verdict_back(float div, float sn, float M, float hi_p, float lo_m) {
    const float sp = sn + M, sm = sn - M;
    const float w = cln_min(cln_min(-div, -fmaf(hi_p, div, sm)), fmaf(lo_m, div, sp));
    return (cln_bits(w) >> 31) == 0u;
}
/* the (div, sn) the sweep computes for an entry: cl < 3: {q, n_a}; cl == 3: {n, k} */
static void entry_eval(int cl, const float* e, v3 o, v3 d, float* div, float* sn) {
    if (cl < 3) {
        const float oa = cl == 0 ? o.x : (cl == 1 ? o.y : o.z), da = cl == 0 ? d.x : (cl == 1 ? d.y : d.z);
        *div = e[1] * da;
        *sn = e[1] * (oa - e[0]);
    } else {
        *div = cln_dot3(e[0], e[1], e[2], d.x, d.y, d.z);
        *sn = fmaf(e[2], o.z, fmaf(e[1], o.y, fmaf(e[0], o.x, -e[3])));
    }
}
/* the sweep's verdict on one record through the entry the list would hold for it; reversed: through the entry of its reversed twin (back mask) */
static int sweep_keeps(v3 o, v3 d, float cmin, float cmax, float maxt, v3 p0, v3 e1, v3 e2, float G, float H, int reversed) {
    const float o1 = (fabsf(o.x) + fabsf(o.y)) + fabsf(o.z);
    const float hi_p = cln_max(cln_min(cmax, maxt) * 1.00000095367431640625f, 0x1p-100f);
    const float lo_m = cmin * 0.99999904632568359375f;
    const v3 n = v3cross(e2, e1);
    const float k = (float)((double)p0.x * n.x + (double)p0.y * n.y + (double)p0.z * n.z);
    const int cl = plane_class(p0, e1, e2, n);
    const float sg = reversed ? -1.0f : 1.0f;   /* the twin's entry: -n, -k (an axis plane keeps its coordinate) */
    float e[4];
    if (cl < 3) { e[0] = canon0(cl == 0 ? p0.x : (cl == 1 ? p0.y : p0.z)); e[1] = sg * (cl == 0 ? n.x : (cl == 1 ? n.y : n.z)); e[2] = e[3] = 0.0f; }
    else { e[0] = canon0(sg * n.x); e[1] = canon0(sg * n.y); e[2] = canon0(sg * n.z); e[3] = canon0(sg * k); }
    float div, sn;
    entry_eval(cl, e, o, d, &div, &sn);
    const float M = fmaf(G, o1, H);
    return reversed ? verdict_back(div, sn, M, hi_p, lo_m) : verdict_front(div, sn, M, hi_p, lo_m);
}

static inline uint64_t mix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline float urand(uint64_t* s) { return (float)((mix64(s) >> 40) * (1.0 / 16777216.0)); }        /* [0, 1) */
static inline float srand1(uint64_t* s) { return 2.0f * urand(s) - 1.0f; }
static inline float ulps(float x, int k) {   /* k representable steps away from x (x finite, positive or negative) */
    uint32_t b = cln_bits(x);
    if ((b & 0x7fffffffu) == 0u) return k >= 0 ? cln_float((uint32_t)k) : -cln_float((uint32_t)(-k));
    if (b >> 31) b -= (uint32_t)k; else b += (uint32_t)k;
    return cln_float(b);
}

/* out[0] = cases, out[1] = violations (dropped by the sweep, accepted by the reference), out[2] = accepted by the reference,
 * out[3] = rejected by the reference, out[4] = of those dropped by the sweep, out[5] = cases outside the ray guard (skipped) */
/* shrink: the margin constants G and H divided by 2^shrink (0 = the product's).  A margin cut to the size of the roundings it must cover
 * makes the violations appear: how the test shows that it can see them. */
void oracle_sweep_check(uint64_t seed, uint64_t count, int shrink, uint64_t* out) {
    uint64_t cases = 0, viol = 0, acc = 0, rej = 0, dropped = 0, skipped = 0;
#pragma omp parallel for reduction(+ : cases, viol, acc, rej, dropped, skipped) schedule(static)
    for (uint64_t it = 0; it < count; ++it) {
        uint64_t s = seed * 0xD1B54A32D192ED03ull + it * 0x9E3779B97F4A7C15ull + 1u;
        /* geometry scale: forty octaves around 1 (inside the guard windows of the optimistic kernel: |coordinate| <= 2^20) */
        const float scale = ldexpf(1.0f, (int)(mix64(&s) % 41u) - 20);
        const int shape = (int)(mix64(&s) % 4u);
        v3 p0 = {srand1(&s), srand1(&s), srand1(&s)}, p1, p2;
        if (shape == 0) {          /* generic */
            p1.x = srand1(&s); p1.y = srand1(&s); p1.z = srand1(&s);
            p2.x = srand1(&s); p2.y = srand1(&s); p2.z = srand1(&s);
        } else if (shape == 1) {   /* axis-aligned quad half on a 1/64 lattice: exact products, as scenes built from walls have them */
            const int ax = (int)(mix64(&s) % 3u);
            float u[3] = {0, 0, 0}, w[3] = {0, 0, 0};
            u[(ax + 1) % 3] = (float)(1 + mix64(&s) % 63u) / 64.0f * (mix64(&s) & 1u ? 1.0f : -1.0f);
            w[(ax + 2) % 3] = (float)(1 + mix64(&s) % 63u) / 64.0f * (mix64(&s) & 1u ? 1.0f : -1.0f);
            p0.x = roundf(p0.x * 64.0f) / 64.0f; p0.y = roundf(p0.y * 64.0f) / 64.0f; p0.z = roundf(p0.z * 64.0f) / 64.0f;
            p1.x = p0.x + u[0]; p1.y = p0.y + u[1]; p1.z = p0.z + u[2];
            p2.x = p0.x + w[0]; p2.y = p0.y + w[1]; p2.z = p0.z + w[2];
        } else if (shape == 2) {   /* needle: one short edge */
            p1.x = p0.x + 1e-3f * srand1(&s); p1.y = p0.y + 1e-3f * srand1(&s); p1.z = p0.z + 1e-3f * srand1(&s);
            p2.x = srand1(&s); p2.y = srand1(&s); p2.z = srand1(&s);
        } else {                   /* far from the origin: |p0| dominates the edges (cancellation in s . n) */
            p0.x = 30.0f * srand1(&s); p0.y = 30.0f * srand1(&s); p0.z = 30.0f * srand1(&s);
            p1.x = p0.x + 0.3f * srand1(&s); p1.y = p0.y + 0.3f * srand1(&s); p1.z = p0.z + 0.3f * srand1(&s);
            p2.x = p0.x + 0.3f * srand1(&s); p2.y = p0.y + 0.3f * srand1(&s); p2.z = p0.z + 0.3f * srand1(&s);
        }
        p0.x *= scale; p0.y *= scale; p0.z *= scale; p1.x *= scale; p1.y *= scale; p1.z *= scale; p2.x *= scale; p2.y *= scale; p2.z *= scale;
        const v3 e1 = v3sub(p1, p0), e2 = v3sub(p2, p0);
        const v3 n = v3cross(e2, e1);
        float G, H;
        plane_margin(p0, e1, e2, &G, &H);
        G = ldexpf(G, -shrink);
        H = ldexpf(H, -shrink);

        /* the ray: aimed at a point of the triangle (inside, on an edge, just outside), or anywhere */
        v3 o = {scale * 2.0f * srand1(&s), scale * 2.0f * srand1(&s), scale * 2.0f * srand1(&s)};
        const int aim = (int)(mix64(&s) % 4u);
        v3 d;
        if (aim < 3) {
            float b = urand(&s), g = urand(&s) * (1.0f - b);
            if (aim == 1) { b = (mix64(&s) & 1u) ? 0.0f : b; g = (mix64(&s) & 1u) ? 0.0f : 1.0f - b; }      /* on an edge / a corner */
            if (aim == 2) { b = b * 1.2f - 0.1f; g = g * 1.2f - 0.1f; }                                    /* around the outline */
            const v3 P = {p0.x + b * e1.x + g * e2.x, p0.y + b * e1.y + g * e2.y, p0.z + b * e1.z + g * e2.z};
            d = v3sub(P, o);
        } else {
            d.x = srand1(&s); d.y = srand1(&s); d.z = srand1(&s);
        }
        if (mix64(&s) % 8u == 0u) {   /* nearly parallel to the plane: nudge d into it */
            const float nn = v3dot(n, n);
            if (nn > 0.0f) { const float f = v3dot(n, d) / nn * (1.0f - 1e-5f * urand(&s)); d.x -= f * n.x; d.y -= f * n.y; d.z -= f * n.z; }
        }
        { const float inv = 1.0f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z); d.x *= inv; d.y *= inv; d.z *= inv; }   /* any unit-ish direction will do */
        if (mix64(&s) & 1u) { d.x = -d.x; d.y = -d.y; d.z = -d.z; }   /* either facing */
        /* the optimistic kernel's ray guard (pt_trace.hpp ray_guard): outside it a sample is deferred, the sweep's word is not used */
        {
            int ok = 1;
            const float dd[3] = {d.x, d.y, d.z}, oo[3] = {o.x, o.y, o.z};
            for (int q = 0; q < 3; ++q) {
                const float ad = fabsf(dd[q]), ao = fabsf(oo[q]);
                ok = ok && ad >= 0x1p-40f && ad <= 0x1p40f && (ao == 0.0f || (ao >= 0x1p-30f && ao <= 0x1p20f));
            }
            /* ... and the geometry side (GridArgs::fast_ok, k_prepTriangles): vertices and edges within 2^21, plane-normal components zero or
             * in [2^-40, 2^40]; a set outside it runs the exact kernel, without the sweep */
            const float gg[9] = {p0.x, p0.y, p0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z}, nn3[3] = {n.x, n.y, n.z};
            for (int q = 0; q < 9; ++q) ok = ok && fabsf(gg[q]) <= 2097152.0f;
            for (int q = 0; q < 3; ++q) { const float an = fabsf(nn3[q]); ok = ok && (an == 0.0f || (an >= 0x1p-40f && an <= 0x1p40f)); }
            if (!ok) { skipped++; continue; }
        }

        /* the window: on and around the reference's own t, or anywhere */
        const float t0 = ref_t(o, d, p0, e1, e2);
        float cmin, cmax, maxt;
        const int win = (int)(mix64(&s) % 4u);
        if (win < 3 && t0 == t0 && t0 > 0.0f) {
            const int a = (int)(mix64(&s) % 7u) - 3, b = (int)(mix64(&s) % 7u) - 3, c = (int)(mix64(&s) % 7u) - 3;
            cmin = win == 0 ? ulps(t0, a) : (win == 1 ? 0.0f : t0 * urand(&s));
            cmax = win == 1 ? ulps(t0, b) : (win == 0 ? t0 * (1.0f + urand(&s)) : ulps(t0, b));
            maxt = win == 2 ? ulps(t0, c) : ((mix64(&s) & 1u) ? INFINITY : ulps(t0, c + 1));
            if (cmin < 0.0f) cmin = 0.0f;
        } else {
            cmin = (mix64(&s) & 1u) ? 0.0f : scale * 4.0f * urand(&s);
            cmax = cmin + scale * 4.0f * urand(&s);
            maxt = (mix64(&s) & 1u) ? INFINITY : scale * 6.0f * urand(&s);
        }
        if (!(cmin <= cmax)) continue;   /* the box test already said no (bh.v) */

        float t;
        const int accept = ref_test(o, d, cmin, cmax, maxt, p0, e1, e2, &t);
        const int keep = sweep_keeps(o, d, cmin, cmax, maxt, p0, e1, e2, G, H, (int)(mix64(&s) & 1u));   /* through its own entry, or its reversed twin's */
        cases++;
        if (accept) { acc++; if (!keep) viol++; }
        else { rej++; if (!keep) dropped++; }
    }
    out[0] = cases; out[1] = viol; out[2] = acc; out[3] = rej; out[4] = dropped; out[5] = skipped;
}

/* ---- the layout: k_planeList (pt_kernels_fused.hip) word for word.  prep: count records {p0.xyz, n.x} {e1.xyz, n.y} {e2.xyz, n.z} as
 * k_prepTriangles leaves them; out: the plane list, 16 + 16 * groups words (at most 16 + 4 * 320 words for 96 records); returns the words written */
uint32_t oracle_plane_list(const float* prep, uint32_t count, uint32_t* out) {
    uint32_t* hdr = out;
    uint32_t* ent = out + 16;
    uint32_t groups = 0;
    for (uint32_t c = 0; c < 4u; ++c) {
        const uint32_t lo = c * 32u, hi = lo + 32u < count ? lo + 32u : count;
        hdr[c] = 0u; hdr[4u + c] = groups; hdr[8u + c] = 0u; hdr[12u + c] = 0u;
        if (lo >= count) continue;
        float gmax = 0.0f, hmax = 0.0f;
        uint32_t cls[32], key[32][4];
        for (uint32_t i = lo; i < hi; ++i) {
            const float* r = prep + 12u * i;
            const v3 p0 = {r[0], r[1], r[2]}, e1 = {r[4], r[5], r[6]}, e2 = {r[8], r[9], r[10]}, n = {r[3], r[7], r[11]};
            const float k = (float)((double)p0.x * n.x + (double)p0.y * n.y + (double)p0.z * n.z);
            float G, H;
            plane_margin(p0, e1, e2, &G, &H);
            gmax = fmaxf(gmax, G); hmax = fmaxf(hmax, H);
            const int cl = plane_class(p0, e1, e2, n);
            cls[i - lo] = (uint32_t)cl;
            const float nn[3] = {n.x, n.y, n.z}, pp[3] = {p0.x, p0.y, p0.z};
            if (cl < 3) { key[i - lo][0] = cln_bits(canon0(pp[cl])); key[i - lo][1] = cln_bits(nn[cl]); key[i - lo][2] = key[i - lo][3] = 0u; }
            else { key[i - lo][0] = cln_bits(canon0(n.x)); key[i - lo][1] = cln_bits(canon0(n.y)); key[i - lo][2] = cln_bits(canon0(n.z)); key[i - lo][3] = cln_bits(canon0(k)); }
        }
        hdr[8u + c] = cln_bits(gmax); hdr[12u + c] = cln_bits(hmax);
        uint32_t packed = 0u;
        for (uint32_t cl = 0; cl < 4u; ++cl) {
            const uint32_t per = cl < 3u ? 4u : 2u, words = cl < 3u ? 4u : 8u;
            uint32_t n_ent = 0, done = 0u;
            uint32_t* base = ent + 16u * groups;
            for (uint32_t i = 0; i < hi - lo; ++i) {
                if (cls[i] != cl || (done >> i & 1u)) continue;
                uint32_t mask = 0u, back = 0u;
                for (uint32_t j = i; j < hi - lo; ++j) {
                    if (cls[j] != cl || (done >> j & 1u)) continue;
                    int same = 1, rev = 1;
                    for (uint32_t w = 0; w < 4u; ++w) {
                        const uint32_t a = key[i][w], b = key[j][w];
                        same = same && a == b;
                        const int coord = cl < 3u && w == 0u;
                        rev = rev && ((coord || a == 0u) ? a == b : (a ^ 0x80000000u) == b);
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
            for (uint32_t w = words * n_ent; w < 16u * g; ++w) base[w] = 0u;
            packed |= g << (8u * cl);
            groups += g;
        }
        hdr[c] = packed;
    }
    return 16u + 16u * groups;
}

/* trace_cell1's sweep over the list for one ray: cand[c] = the candidate word of chunk c (record 32 c + j at bit 31 - j) */
void oracle_sweep_words(const uint32_t* list, uint32_t count, const float* o3, const float* d3, float cmin, float cmax, float maxt, uint32_t* cand) {
    const v3 o = {o3[0], o3[1], o3[2]}, d = {d3[0], d3[1], d3[2]};
    const float o1 = (fabsf(o.x) + fabsf(o.y)) + fabsf(o.z);
    const float hi_p = cln_max(cln_min(cmax, maxt) * 1.00000095367431640625f, 0x1p-100f);
    const float lo_m = cmin * 0.99999904632568359375f;
    for (uint32_t c0 = 0, c = 0; c0 < count; c0 += 32u, ++c) {
        const uint32_t groups = list[c];
        const uint32_t* g = list + 16 + 16u * list[4u + c];
        const float Mu = fmaf(cln_float(list[8u + c]), o1, cln_float(list[12u + c]));
        uint32_t w = 0u;
        for (uint32_t cl = 0; cl < 4u; ++cl) {
            const uint32_t ng = (groups >> (8u * cl)) & 255u, per = cl < 3u ? 4u : 2u, words = cl < 3u ? 4u : 8u;
            for (uint32_t i = 0; i < ng; ++i, g += 16) {
                for (uint32_t k = 0; k < per; ++k) {
                    const uint32_t* e = g + words * k;
                    const uint32_t mask = cl < 3u ? e[2] : e[4], back = cl < 3u ? e[3] : e[5];
                    if ((mask | back) == 0u) continue;
                    float ef[4] = {cln_float(e[0]), cln_float(e[1]), cln_float(e[2]), cln_float(e[3])};
                    float div, sn;
                    entry_eval((int)cl, ef, o, d, &div, &sn);
                    if (verdict_front(div, sn, Mu, hi_p, lo_m)) w |= mask;
                    if (back != 0u && verdict_back(div, sn, Mu, hi_p, lo_m)) w |= back;
                }
            }
        }
        cand[c] = w;
    }
}

/* the reference's verdict on record i of `prep` for the same ray and window (what a dropped candidate must not be) */
int oracle_sweep_ref_accepts(const float* prep, uint32_t i, const float* o3, const float* d3, float cmin, float cmax, float maxt) {
    const float* r = prep + 12u * i;
    const v3 o = {o3[0], o3[1], o3[2]}, d = {d3[0], d3[1], d3[2]}, p0 = {r[0], r[1], r[2]}, e1 = {r[4], r[5], r[6]}, e2 = {r[8], r[9], r[10]};
    float t;
    return ref_test(o, d, cmin, cmax, maxt, p0, e1, e2, &t);
}
