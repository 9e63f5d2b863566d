// pt_device.hpp -- device building blocks of the Assign10 path (gfx950): vectors, layouts, RNG, camera, ray/box, the DDA
// axis set-up, lights, shading and bounce sampling.  Primitive tests and grid traversal live in pt_trace.hpp.
//
// Shared by both kernel families:
//   * the reference-shaped granular kernels (pt_kernels_granular.hip), drop-in
//     for the fourteen `__kernel`s of A10 code.cl behind the WebCL-shaped API;
//   * the fused per-ray pass kernel (pt_kernels_fused.hip).
// "A10 code.cl:NNN" cites the reference lines whose behaviour a block reproduces.
// Evaluation order and fusion are part of the contract (pt_numerics.hpp): every expression is spelled in the order OpenCL C
// evaluates the reference's, with fma3 / cl_fma exactly where the OpenCL front end contracts the reference's text.
#pragma once
#include "pt_numerics.hpp"

namespace pt {

#define PT_INF_ (__builtin_inff())
struct f3 { float x, y, z; };

PT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 scl3(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
PT_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// s*a + c, one rounding per component: the sites the OpenCL front end contracts (A10 code.cl:87, 111, 190, 410, 574, 659, 666)
PT_DEV f3 fma3(float s, f3 a, f3 c) { return mk3(cl_fma(s, a.x, c.x), cl_fma(s, a.y, c.y), cl_fma(s, a.z, c.z)); }
// dot, cross, length, normalize: AMD's OpenCL library definitions (opencl.bc), see pt_numerics.hpp
PT_DEV float dot3(f3 a, f3 b) { return cl_fma(a.z, b.z, cl_fma(a.y, b.y, a.x * b.x)); }
PT_DEV f3 cross3(f3 a, f3 b) {
    return mk3(cl_fma(a.y, b.z, -(a.z * b.y)), cl_fma(a.z, b.x, -(a.x * b.z)), cl_fma(a.x, b.y, -(a.y * b.x)));
}
// The library's corner cases (dot below 2^-126, infinite, NaN, zero vector) are rare-lane overrides of the plain result: written with
// the plain path first, the compiler keeps it three or four instructions long (as two branches of an if it merged the tails and every
// lane ran the rescaling selects).  One unsigned compare tells the cases apart: dot(v, v) is never negative, so "normal and finite" is
// bits - 2^-126's bits < inf's bits - 2^-126's bits (zero wraps to the top; inf and NaN sit at or above the bound).
PT_DEV bool dot_plain(float d) { return __float_as_uint(d) - 0x00800000u < 0x7F000000u; }
PT_DEV float len3(f3 a) {
    const float d = dot3(a, a);
    float r = __builtin_amdgcn_sqrtf(d);
    if (__builtin_expect(!dot_plain(d), 0)) {
        if (d < 0x1p-126f) { const f3 b = scl3(0x1p+86f, a); r = cl_sqrt_approx(dot3(b, b)) * 0x1p-86f; }
        else if (d == PT_INF_) { const f3 b = scl3(0x1p-66f, a); r = cl_sqrt_approx(dot3(b, b)) * 0x1p+66f; }
    }
    return r;
}
PT_DEV f3 norm3(f3 a) {
    float d = dot3(a, a);
    f3 r = scl3(__builtin_amdgcn_rsqf(d), a);
    if (__builtin_expect(!dot_plain(d), 0)) {   // the library's corner cases, in its order
        if (a.x == 0.0f && a.y == 0.0f && a.z == 0.0f) return a;
        if (d < 0x1p-126f) { a = scl3(0x1p+86f, a); d = dot3(a, a); }
        else if (d == PT_INF_) {
            a = scl3(0x1p-66f, a); d = dot3(a, a);
            if (d == PT_INF_) {
                a = mk3(__builtin_copysignf(__builtin_isinf(a.x) ? 1.0f : 0.0f, a.x), __builtin_copysignf(__builtin_isinf(a.y) ? 1.0f : 0.0f, a.y),
                        __builtin_copysignf(__builtin_isinf(a.z) ? 1.0f : 0.0f, a.z));
                d = dot3(a, a);
            }
        }
        r = scl3(cl_rsqrt(d), a);
    }
    return r;
}
PT_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
PT_DEV f3 ld3(const float4& v) { return mk3(v.x, v.y, v.z); }

#define PT_INF PT_INF_
#define PT_PI_4 0.785398163397448309616f
#define PT_PI_2 1.57079632679489661923f

// ---- AoS layouts of the reference interface (measured from the compiled reference:
// Ray 48 B, Poi 64 B) ------------------------------------------
struct alignas(16) RayAoS { float ox, oy, oz, _p0, dx, dy, dz, _p1, mint, maxt, _p2, _p3; };
struct alignas(16) PoiAoS { float px, py, pz, _p0, nx, ny, nz, _p1, ax, ay, az, _p2; int32_t matId; int32_t _p3[3]; };
static_assert(sizeof(RayAoS) == 48 && sizeof(PoiAoS) == 64, "reference struct sizes");

struct Ray { f3 o, d; float mint, maxt; };
struct Box { f3 lo, hi; };                  // AABB, host packing (min,1,max,1): A10 code.js:610-621
struct Cam { f3 eye, U, V, W; float width, height; uint32_t cols, rows; };

struct Box8 { float v[8]; };                // by-value kernel arguments
struct F16 { float v[16]; };

PT_DEV Box mk_box(const Box8& b) { Box r; r.lo = ld3(b.v); r.hi = ld3(b.v + 4); return r; }
// A10 code.cl:73-84
PT_DEV Cam mk_cam(const F16& f) {
    Cam c;
    c.eye = ld3(f.v); c.U = ld3(f.v + 3); c.V = ld3(f.v + 6); c.W = ld3(f.v + 9);
    c.width = f.v[12]; c.height = f.v[13];
    c.cols = f2u_uniform(f.v[14]); c.rows = f2u_uniform(f.v[15]);
    return c;
}

// ---- RNG: A10 code.cl:420-434 --------------------------------------------------------
// seed' = ((long)(int32)(seed*16807)) % (2^31-1): int32 wrap first, then C's truncating
// remainder.  For an int32 x, x % (2^31-1) is x itself except at three values.
PT_DEV int32_t lcg_next(int32_t s) {
    int32_t w = (int32_t)((uint32_t)s * 16807u);
    // the three values are one unsigned range [0x7FFFFFFF, 0x80000001] = {INT32_MAX, INT32_MIN, -INT32_MAX}: one compare, a rare fix-up
    if (__builtin_expect((uint32_t)w - 0x7FFFFFFFu <= 2u, 0)) w = (w == INT32_MIN) ? -1 : 0;
    return w;
}
PT_DEV float lcg_float(int32_t s) { return cl_fabs((float)s * 4.656612873077392578125e-10f); }  // * 2^-31
PT_DEV float get_rand(int32_t& seed) { seed = lcg_next(seed); return lcg_float(seed); }

// ---- camera: A10 code.cl:108-119, 143-197 ------------------------------------------
PT_DEV void concentric(float inx, float iny, float& ox, float& oy) {
    if (inx == 0.0f && iny == 0.0f) { ox = inx; oy = iny; return; }
    float a = cl_fma(2.0f, inx, -1.0f);        // code.cl:152-153, contracted
    float b = cl_fma(2.0f, iny, -1.0f);
    const bool top = (a * a) > (b * b);
    float radius = top ? (1.0f * a) : (1.0f * b);
    const float num = top ? b : a, den = top ? a : b;
    float q = num / den;   // code.cl:158, 163: b / a or a / b -- one division on the selected operands (written as a select of two
                           // quotients the compiler evaluates both: eleven instructions per call for nothing)
    float phi = top ? (PT_PI_4 * q) : cl_fma(-PT_PI_4, q, PT_PI_2);   // code.cl:164: c - a*b contracts to fma(-a, b, c)
    float s, c;
    cl_sincos(phi, s, c);
    ox = c * radius;
    oy = s * radius;
}

PT_DEV f3 focal_point(const Cam& c, float col, float row, float focal_length) {
    float sx = (-0.5f + (col + 0.5f) / (float)c.cols) * c.width;
    float sy = (0.5f - (row + 0.5f) / (float)c.rows) * c.height;
    f3 cop = add3(fma3(sx, c.U, scl3(sy, c.V)), scl3(-1.0f, c.W));   // code.cl:111-113: fma(sx, U, sy*V), then + (-W) (exact either way)
    f3 d = norm3(cop);
    f3 o = c.eye;
    f3 pip = add3(c.eye, scl3(-1.0f, scl3(focal_length, c.W)));       // code.cl:176: the product by -1 is exact, fused or not
    float pd = -dot3(pip, c.W);
    float t = -(dot3(o, c.W) + pd) / dot3(d, c.W);
    return fma3(t, d, o);                                              // getPoint, code.cl:87
}

PT_DEV Ray thin_lens_ray(const Cam& c, f3 fp, float lens_rad, float cx, float cy) {
    Ray r;
    float dx, dy;
    concentric(cx, cy, dx, dy);
    dx = dx * lens_rad;
    dy = dy * lens_rad;
    r.o = fma3(dy, c.V, fma3(dx, c.U, c.eye));                        // code.cl:190
    r.d = norm3(sub3(fp, r.o));
    r.mint = 0.0f;
    r.maxt = PT_INF;
    return r;
}

// ---- ray / box: A10 code.cl:335-389 ---------------------------------------------------
struct BoxHit { float tmin, tmax; bool v; float tfx, tfy, tfz; };   // tf*: per-axis exit parameter (the slab's far t)

PT_DEV bool slab1(float lo, float hi, float o, float d, BoxHit& h, float& tfar) {
    float t0 = (lo - o) / d;
    float t1 = (hi - o) / d;
    const bool neg = d < 0;
    float tn = neg ? t1 : t0;
    float tf = neg ? t0 : t1;
    tfar = tf;
    h.tmin = cl_max(tn, h.tmin);
    h.tmax = cl_min(tf, h.tmax);
    return !(h.tmin > h.tmax);
}
PT_DEV BoxHit inter_aabb(const Ray& r, const Box& b) {
    // The reference returns at the first slab with tmin > tmax (code.cl:351-354, 366-369,
    // 381-384); the later slabs only refine values nobody reads after a failure, so all
    // three are evaluated here and the verdicts and-ed: same `v`, same tmin/tmax when v.
    BoxHit h;
    h.tmin = 0.0f;
    h.tmax = PT_INF;
    const bool okx = slab1(b.lo.x, b.hi.x, r.o.x, r.d.x, h, h.tfx);
    const bool oky = slab1(b.lo.y, b.hi.y, r.o.y, r.d.y, h, h.tfy);
    const bool okz = slab1(b.lo.z, b.hi.z, r.o.z, r.d.z, h, h.tfz);
    h.v = okx && oky && okz;
    return h;
}
// primary-ray clip of initTrace (code.cl:494-501): miss -> mint = maxt ("dead ray")
PT_DEV void clip_to(Ray& r, const Box& b) {
    BoxHit h = inter_aabb(r, b);
    const float far_ = r.maxt;
    r.mint = h.v ? h.tmin : far_;
    r.maxt = h.v ? h.tmax : far_;
}

// ---- 3-D uniform grid, 3-axis DDA: the traversal the reference repeats at
// A10 code.cl:694-786, 822-919, 957-1054, 1090-1183, 1213-1310 -------------------------
struct Axis { int slab, dslab, limit; float dt, tnext; };

PT_DEV Axis axis_setup(float o, float d, float tmin, float lo, float hi, uint32_t n) {
    Axis a;
    float x = cl_fma(tmin, d, o);                                      // code.cl:698
    float delta = (hi - lo) / (float)n;
    a.slab = f2i((x - lo) / delta);
    if (a.slab < 0) a.slab = 0;
    if ((uint32_t)a.slab >= n) a.slab = (int)(n - 1u);
    const bool fwd = d >= 0;
    a.dslab = fwd ? 1 : -1;
    a.limit = fwd ? (int)n : -1;
    a.dt = delta / cl_fabs(d);
    float xnext = cl_fma((float)(a.slab + (fwd ? 1 : 0)), delta, lo);  // code.cl:706
    a.tnext = (xnext - o) / d;
    return a;
}

enum PrimKind { SPHERES = 0, TRIANGLES = 1 };

// A path vertex ("point of intersection"), code.cl:57-62
struct Poi { f3 p, n, atte; int32_t matId; };

// ---- lights ------------------------------------------------------------------------------
// code.cl:391-403 + 600-629.  Returns true when the disk emitter is seen before the surface.
PT_DEV bool light_visible(const Ray& ray, f3 lpos, f3 lnor, float radius) {
    float den = dot3(ray.d, lnor);
    if (den == 0.0f) return false;
    float num = dot3(sub3(lpos, ray.o), lnor);
    if (num == 0.0f) return false;
    float t = num / den;
    f3 p = fma3(t, ray.d, ray.o);
    if (len3(sub3(p, lpos)) > radius) return false;
    return !(t >= ray.maxt);
}

// code.cl:121-129 + 631-673: offset origin, concentric sample on the disk light, ray to it
PT_DEV Ray shadow_ray(const Poi& poi, f3 lpos0, f3 T, f3 B, float radius, int32_t& seed) {
    f3 p = fma3(0.001f, poi.n, poi.p);                                 // code.cl:659
    float x = get_rand(seed);
    float y = get_rand(seed);
    concentric(x, y, x, y);
    x = x * radius;
    y = y * radius;
    f3 lpos = add3(lpos0, fma3(x, T, scl3(y, B)));                     // code.cl:666
    f3 to = sub3(lpos, p);
    Ray r;
    r.o = p;
    r.d = norm3(to);
    r.mint = 0.0f;
    r.maxt = len3(to);
    return r;
}

// code.cl:1323-1364 minus the memory traffic: returns material*atte*shade and scales atte
PT_DEV f3 shade_vertex(Poi& poi, const Ray& sh, f3 color, f3 lpos, f3 lnor, f3 es, float area) {
    f3 shade = mk3(0.0f, 0.0f, 0.0f);
    if (sh.maxt != sh.mint) {
        float r = len3(sub3(poi.p, lpos));
        float cosx = cl_clamp(dot3(sh.d, poi.n), 0.0f, 1.0f);
        float cosy = cl_clamp(dot3(neg3(sh.d), lnor), 0.0f, 1.0f);
        shade = scl3(area * ((cosx * cosy) / (r * r)), es);
    }
    f3 atte = poi.atte;
    poi.atte = mul3(atte, color);
    return mul3(mul3(color, atte), shade);
}

// code.cl:545-579: cosine-ish hemisphere direction about the vertex normal
PT_DEV Ray bounce_ray(const Poi& poi, int32_t& seed) {
    f3 N = mk3(cl_fabs(poi.n.x), cl_fabs(poi.n.y), cl_fabs(poi.n.z));
    f3 B = poi.n;
    float nmin = cl_min(cl_min(N.x, N.y), N.z);
    if (N.x == nmin) B.x = 1.0f;
    else if (N.y == nmin) B.y = 1.0f;
    else B.z = 1.0f;
    N = poi.n;
    B = norm3(B);
    f3 T = cross3(B, N);
    B = cross3(N, T);
    float sx = get_rand(seed);
    float sy = get_rand(seed);
    concentric(sx, sy, sx, sy);
    float sz = cl_sqrt(cl_max(0.0f, cl_fma(-sy, sy, cl_fma(-sx, sx, 1.0f))));   // code.cl:568
    Ray r;
    r.o = poi.p;
    r.d = norm3(fma3(sz, N, fma3(sx, T, scl3(sy, B))));                          // code.cl:574
    r.mint = 0.0f;
    r.maxt = PT_INF;
    return r;
}

}  // namespace pt
