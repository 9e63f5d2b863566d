/*
 * oracle/pt_oracle.c -- TEST INFRASTRUCTURE (checker only; see pt_oracle.h).
 *
 * CPU restatement, in plain C, of the Assign10 path-tracing kernels of
 * eaymerich/2015-RayTracing.  Every function cites the reference lines it
 * follows ("A10 code.cl:NNN" = Assign10-Path_Tracing/code.cl).  Arithmetic
 * follows oracle/cl_numerics.h: the reference as AMD's OpenCL toolchain builds it for
 * gfx950.  That means (a) the built-ins are AMD's (fused dot / cross, hardware
 * rsq / sqrt in normalize / length, v_min / v_max / v_med3) and (b) a*b+c is FUSED
 * wherever the OpenCL front end contracts it: a multiplication that feeds an
 * addition or subtraction inside one expression (the left operand first when both
 * are products).  The sites in the reference's text (A10 code.cl:87, 111, 152, 153,
 * 164, 190, 410, 568, 574, 659, 666, 698-732 and its four copies; oracle/README.md
 * has the list and how it was obtained) are spelled fma3 / cln_fma below; everything
 * else is separate IEEE operations, component-wise, in the order OpenCL C evaluates
 * them.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math: fusion is
 * explicit).
 */
#include "pt_oracle.h"
#include "cl_numerics.h"

#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- flop accounting ---------------------------------------- */
#ifdef PTO_COUNT_FLOPS
static unsigned long long g_flops;
#define FL(n) (g_flops += (unsigned long long)(n))
unsigned long long oracle_flops_get(void) { return g_flops; }
void oracle_flops_reset(void) { g_flops = 0; }
/* grid-walk statistics of the triangle walks over grids with n > 1 (profiles/walk_stats.py: how much of the walk's work is a triangle tested
 * again in a later cell).  [0] walks, [1] cells visited, [2] cells with a list, [3] tests, [4] tests of a triangle this ray already tested in
 * an earlier cell of the walk, [5] ... whose earlier test was rejected for a reason no cell changes (facing, barycentrics), [6] ... rejected
 * by the cell's window alone, [7] walks that end with a hit, [8] tests the reference makes in cells that start beyond the ray's end (the
 * product's walks stop there). */
static unsigned long long g_walk[16];
static int g_last_reject;   /* of the last inter_triangle: 0 accepted, 1 facing, 2 barycentrics, 3 window */
#define REJ(k) (g_last_reject = (k))
void oracle_walk_stats_get(unsigned long long* out) { for (int i = 0; i < 16; ++i) out[i] = g_walk[i]; }
void oracle_walk_stats_reset(void) { for (int i = 0; i < 16; ++i) g_walk[i] = 0; }
#else
#define FL(n) ((void)0)
#define REJ(k) ((void)0)
unsigned long long oracle_flops_get(void) { return 0; }
void oracle_flops_reset(void) {}
#endif

int oracle_num_threads(void) {
#if defined(_OPENMP) && !defined(PTO_COUNT_FLOPS)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_threads(int n) {
#if defined(_OPENMP) && !defined(PTO_COUNT_FLOPS)
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#if defined(_OPENMP) && !defined(PTO_COUNT_FLOPS)
#define PAR_FOR _Pragma("omp parallel for schedule(static, 4096)")
#define PAR_ROWS _Pragma("omp parallel for schedule(dynamic, 1)")
#else
#define PAR_FOR
#define PAR_ROWS
#endif

/* ---- small vector layer -------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { FL(3); return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { FL(3); return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { FL(3); return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl(float s, v3 a) { FL(3); return V(s * a.x, s * a.y, s * a.z); }
/* s*a + c with ONE rounding per component: where the OpenCL front end contracts (see the header comment) */
static inline v3 fma3(float s, v3 a, v3 c) { FL(6); return V(cln_fma(s, a.x, c.x), cln_fma(s, a.y, c.y), cln_fma(s, a.z, c.z)); }
/* the geometric built-ins: AMD's definitions (cl_numerics.h) */
static inline float dot3(v3 a, v3 b) { FL(5); return cln_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
static inline v3 cross3(v3 a, v3 b) {
    FL(9);
    const float A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z};
    float r[3];
    cln_cross3(A, B, r);
    return V(r[0], r[1], r[2]);
}
static inline float len3(v3 a) { FL(6); return cln_length3(a.x, a.y, a.z); }
static inline v3 norm3(v3 a) {
    FL(9);
    const float A[3] = {a.x, a.y, a.z};
    float r[3];
    cln_normalize3(A, r);
    return V(r[0], r[1], r[2]);
}
/* getPoint, A10 code.cl:86-88: r.o + t * r.d, contracted */
static inline v3 get_point(v3 o, float t, v3 d) { return fma3(t, d, o); }

#define PT_INF (__builtin_inff())
#define PI_4_F 0.785398163397448309616f
#define PI_2_F 1.57079632679489661923f

typedef struct { v3 o, d; float mint, maxt; } ray_t;
typedef struct { v3 pmin, pmax; } box_t;
typedef struct { v3 eye, U, V, W; float width, height; uint32_t cols, rows; } cam_t;

static inline ray_t ld_ray(const pto_ray* r) { ray_t q = {ld3(r->o), ld3(r->d), r->mint, r->maxt}; return q; }
static inline void st_ray(pto_ray* r, ray_t q) { st3(r->o, q.o); st3(r->d, q.d); r->mint = q.mint; r->maxt = q.maxt; }
static inline box_t ld_box(const float* b8) { box_t b = {ld3(b8), ld3(b8 + 4)}; return b; }

/* A10 code.cl:73-84 floatToCamera */
static cam_t ld_cam(const float* f) {
    cam_t c;
    c.eye = ld3(f); c.U = ld3(f + 3); c.V = ld3(f + 6); c.W = ld3(f + 9);
    c.width = f[12]; c.height = f[13];
    c.cols = cln_f2u(f[14]); c.rows = cln_f2u(f[15]);
    return c;
}

/* ---- RNG: A10 code.cl:420-434 -------------------------------------------- */
/* seed' = ((long)(int32)(seed * 16807)) % 2147483647 : the product wraps in
 * int32 BEFORE widening, the remainder is C's truncating one (sign follows
 * the dividend), so states go negative. */
static inline int32_t lcg_next(int32_t s) {
    int32_t w = (int32_t)((uint32_t)s * 16807u);
    return (int32_t)((int64_t)w % 2147483647LL);
}
int oracle_a10_rand(int* seed) { *seed = lcg_next(*seed); return *seed; }

static inline float get_rand(int* slot) {
    const float im = 1.0f / 2147483647.0f; /* == 2^-31 after rounding the divisor */
    int32_t s = lcg_next(*slot);
    *slot = s;
    FL(1);
    return cln_fabs((float)s * im);
}

/* ---- camera: A10 code.cl:108-119, 143-197 -------------------------------- */
static ray_t get_ray(const cam_t* c, float col, float row) {
    FL(8);
    float sx = (-0.5f + (col + 0.5f) / (float)c->cols) * c->width;
    float sy = (0.5f - (row + 0.5f) / (float)c->rows) * c->height;
    /* sx*U + sy*V + (-1)*W: the first sum takes its LEFT product fused, fma(sx, U, sy*V); the second is fma(-1, W, .) = exact */
    v3 cop = add(fma3(sx, c->U, scl(sy, c->V)), scl(-1.0f, c->W));
    ray_t r;
    r.d = norm3(cop);
    r.o = c->eye;
    r.mint = 0.0f;
    r.maxt = PT_INF;
    return r;
}

/* Shirley-Chiu concentric map as the reference spells it (code.cl:143-172) */
static void concentric(float inx, float iny, float* ox, float* oy) {
    if (inx == 0.0f && iny == 0.0f) { *ox = inx; *oy = iny; return; }
    float phi, radius;
    FL(6);
    float a = cln_fma(2.0f, inx, -1.0f);     /* code.cl:152-153, contracted */
    float b = cln_fma(2.0f, iny, -1.0f);
    if ((a * a) > (b * b)) {
        FL(3);
        radius = 1.0f * a;
        phi = PI_4_F * (b / a);
    } else {
        FL(4);
        radius = 1.0f * b;
        phi = cln_fma(-PI_4_F, a / b, PI_2_F);   /* code.cl:164: c - a*b contracts to fma(-a, b, c) */
    }
    float s, c;
    cln_sincos(phi, &s, &c);
    FL(4);
    *ox = c * radius;
    *oy = s * radius;
}

/* code.cl:174-181 */
static v3 focal_point(const cam_t* c, float col, float row, float focal_length) {
    ray_t r = get_ray(c, col, row);
    v3 pip = add(c->eye, scl(-1.0f, scl(focal_length, c->W)));
    v3 N = c->W;
    FL(2);
    float d = -dot3(pip, N);
    float t = -(dot3(r.o, N) + d) / dot3(r.d, N);
    return get_point(r.o, t, r.d);
}

/* code.cl:183-197 */
static ray_t thin_lens_ray(const cam_t* c, v3 fp, float lens_rad, float cx, float cy) {
    ray_t r;
    r.mint = 0.0f;
    r.maxt = PT_INF;
    float dx, dy;
    concentric(cx, cy, &dx, &dy);
    FL(2);
    dx = dx * lens_rad;
    dy = dy * lens_rad;
    r.o = fma3(dy, c->V, fma3(dx, c->U, c->eye));   /* code.cl:190, both sums contracted */
    r.d = norm3(sub(fp, r.o));
    return r;
}

/* ---- ray / box: A10 code.cl:335-389 -------------------------------------- */
typedef struct { float tmin, tmax; int v; } boxhit_t;

static boxhit_t inter_aabb(const ray_t* r, const box_t* b) {
    boxhit_t h;
    h.tmin = 0.0f;
    h.tmax = PT_INF;
    h.v = 0;
    const float lo[3] = {b->pmin.x, b->pmin.y, b->pmin.z};
    const float hi[3] = {b->pmax.x, b->pmax.y, b->pmax.z};
    const float o[3] = {r->o.x, r->o.y, r->o.z};
    const float d[3] = {r->d.x, r->d.y, r->d.z};
    for (int k = 0; k < 3; ++k) {
        FL(4);
        float t0 = (lo[k] - o[k]) / d[k];
        float t1 = (hi[k] - o[k]) / d[k];
        if (d[k] < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        h.tmin = cln_max(t0, h.tmin);
        h.tmax = cln_min(t1, h.tmax);
        if (h.tmin > h.tmax) return h;
    }
    h.v = 1;
    return h;
}

/* ---- primitives ---------------------------------------------------------- */
/* code.cl:199-242; sph[3] holds r^2 (host pushes rad*rad, code.js:1602) */
static int inter_sphere(const ray_t* r, const float* sph, float* t_out) {
    v3 omc = sub(r->o, ld3(sph));
    float a = dot3(r->d, r->d);
    FL(6);
    float b = 2.0f * dot3(omc, r->d);
    float c = dot3(omc, omc) - sph[3];
    float dis = cln_mad(-4.0f * c, a, b * b);
    if (dis < 0.0f) { *t_out = PT_INF; return 0; }
    FL(7);
    a = 1.0f / (2.0f * a);
    dis = cln_sqrt(dis);
    float t0 = (-b - dis) * a;
    float t1 = (-b + dis) * a;
    float tmin = cln_fmin(t0, t1);
    float tmax = cln_fmax(t0, t1);
    if (tmin >= r->mint && tmin <= r->maxt) { *t_out = tmin; return 1; }
    if (tmax >= r->mint && tmax <= r->maxt) { *t_out = tmax; return 1; }
    return 0;
}

/* code.cl:250-288: Moeller-Trumbore, single-sided (div <= 0 rejects) */
static int inter_triangle(const ray_t* r, const float* tp /* 3 x float4 */, float* t_out, float* beta_out, float* gamma_out) {
    v3 p0 = ld3(tp), p1 = ld3(tp + 4), p2 = ld3(tp + 8);
    v3 e1 = sub(p1, p0);
    v3 e2 = sub(p2, p0);
    float div = dot3(cross3(e2, e1), r->d);
    if (div <= 0) { REJ(1); return 0; }
    FL(2);
    float idiv = 1.0f / div;
    v3 s = sub(r->o, p0);
    float beta = dot3(cross3(s, r->d), e2) * idiv;
    if (beta < 0.0f || beta > 1.0f) { REJ(2); return 0; }
    FL(1);
    float gamma = dot3(cross3(s, e1), r->d) * idiv;
    FL(1);
    float gb = gamma + beta;
    if (gamma < 0.0f || gb < 0.0f || gb > 1.0f) { REJ(2); return 0; }
    FL(1);
    float t = dot3(cross3(s, e2), e1) * -idiv;
    if (t >= r->mint && t <= r->maxt) { *t_out = t; *beta_out = beta; *gamma_out = gamma; REJ(0); return 1; }
    REJ(3);
    return 0;
}

/* ---- 3-D uniform grid DDA: the text repeated at A10 code.cl:694-786,
 *      822-919, 957-1054, 1090-1183, 1213-1310 ------------------------------ */
typedef struct {
    int slab, dslab, limit;
    float dt, tnext;
} axis_t;

static axis_t axis_setup(float o, float d, float tmin, float lo, float hi, uint32_t n) {
    axis_t a;
    FL(11);
    float x = cln_fma(tmin, d, o);                 /* code.cl:698 */
    float delta = (hi - lo) / (float)n;
    a.slab = cln_f2i((x - lo) / delta);
    if (a.slab < 0) a.slab = 0;
    if ((uint32_t)a.slab >= n) a.slab = (int)(n - 1u);
    a.dslab = (d >= 0) ? 1 : -1;
    a.limit = (d >= 0) ? (int)n : -1;
    a.dt = delta / cln_fabs(d);
    float xnext = cln_fma((float)(a.slab + ((d >= 0) ? 1 : 0)), delta, lo);   /* code.cl:706 */
    a.tnext = (xnext - o) / d;
    return a;
}

enum { PRIM_SPHERE = 0, PRIM_TRIANGLE = 1 };

typedef struct {
    uint32_t idx;   /* UINT_MAX = none */
    float t, beta, gamma;
} champ_t;

/* Walks the grid front to back and stops at the first cell that produced a hit.
 * any_hit: leave the cell scan at the first accepted primitive (shadow kernels). */
static champ_t grid_trace(ray_t ray /* by value: mint/maxt are clobbered per cell */, boxhit_t bh, const box_t* bound,
                          uint32_t n, const uint32_t* cell_off, const float* prims, int kind, int any_hit) {
    axis_t ax = axis_setup(ray.o.x, ray.d.x, bh.tmin, bound->pmin.x, bound->pmax.x, n);
    axis_t ay = axis_setup(ray.o.y, ray.d.y, bh.tmin, bound->pmin.y, bound->pmax.y, n);
    axis_t az = axis_setup(ray.o.z, ray.d.z, bh.tmin, bound->pmin.z, bound->pmax.z, n);
    champ_t ch;
    ch.idx = UINT_MAX;
    ch.t = ray.maxt;
    ch.beta = ch.gamma = 0.0f;
    float t = bh.tmin;
    const uint32_t zs = n * n, ys = n;
#ifdef PTO_COUNT_FLOPS
    const int stats = kind == PRIM_TRIANGLE && n > 1;
    const float ray_end = ray.maxt;
    struct { const float* rec; int rej; } seen[512];
    int n_seen = 0;
    if (stats) g_walk[0]++;
#endif
    for (;;) {
        ray.mint = t;
        ray.maxt = cln_min(cln_min(ax.tnext, ay.tnext), az.tnext);
        uint32_t cell = (uint32_t)az.slab * zs + (uint32_t)ay.slab * ys + (uint32_t)ax.slab;
        uint32_t begin = cell_off[cell], end = cell_off[cell + 1];
#ifdef PTO_COUNT_FLOPS
        if (stats) { g_walk[1]++; if (end > begin) g_walk[2]++; }
#endif
        for (uint32_t i = begin; i < end; ++i) {
            float ti, b = 0.0f, g = 0.0f;
            int hit = (kind == PRIM_SPHERE) ? inter_sphere(&ray, prims + 4u * (size_t)i, &ti)
                                            : inter_triangle(&ray, prims + 12u * (size_t)i, &ti, &b, &g);
#ifdef PTO_COUNT_FLOPS
            if (stats) {
                if (t >= ray_end) g_walk[8]++;
                else {
                    g_walk[3]++;
                    int k;
                    for (k = 0; k < n_seen; ++k)
                        if (memcmp(seen[k].rec, prims + 12u * (size_t)i, 48) == 0) break;
                    if (k < n_seen) {
                        g_walk[4]++;
                        if (seen[k].rej == 1 || seen[k].rej == 2) g_walk[5]++;
                        else if (seen[k].rej == 3) g_walk[6]++;
                        seen[k].rej = g_last_reject;
                    } else if (n_seen < 512) { seen[n_seen].rec = prims + 12u * (size_t)i; seen[n_seen].rej = g_last_reject; ++n_seen; }
                }
            }
#endif
            if (hit && ti < ch.t) {
                ch.t = ti; ch.idx = i; ch.beta = b; ch.gamma = g;
                if (any_hit) break;
            }
        }
#ifdef PTO_COUNT_FLOPS
        if (stats && ch.idx < UINT_MAX) g_walk[7]++;
#endif
        if (ch.idx < UINT_MAX) break;
        t = ray.maxt;
        if (t == ax.tnext) {
            FL(1);
            ax.tnext += ax.dt;
            if (t >= bh.tmax) break;
            ax.slab += ax.dslab;
            if (ax.slab == ax.limit) break;
        } else if (t == ay.tnext) {
            FL(1);
            ay.tnext += ay.dt;
            if (t >= bh.tmax) break;
            ay.slab += ay.dslab;
            if (ay.slab == ay.limit) break;
        } else {
            FL(1);
            az.tnext += az.dt;
            if (t >= bh.tmax) break;
            az.slab += az.dslab;
            if (az.slab == az.limit) break;
        }
    }
    return ch;
}

/* ---- kernels ------------------------------------------------------------- */
unsigned oracle_a10_sizeofRay(void) { return (unsigned)sizeof(pto_ray); } /* code.cl:440-442 */
unsigned oracle_a10_sizeofPoi(void) { return (unsigned)sizeof(pto_poi); } /* code.cl:444-446 */

/* code.cl:448-456 */
void oracle_a10_initAcu(void* acu_, unsigned total, size_t gsz) {
    float* acu = (float*)acu_;
    size_t n = gsz < total ? gsz : total;
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        acu[4 * id + 0] = 0.0f; acu[4 * id + 1] = 0.0f; acu[4 * id + 2] = 0.0f; acu[4 * id + 3] = 0.0f;
    }
}

static void clip_and_store(pto_ray* dst, ray_t ray, const box_t* bound) {
    boxhit_t h = inter_aabb(&ray, bound);
    if (h.v) { ray.mint = h.tmin; ray.maxt = h.tmax; }
    else { ray.mint = ray.maxt; }
    st_ray(dst, ray);
}

/* code.cl:458-543.  2-D NDRange, work-items in row-major order; with rpp == 1
 * the two lens draws come from seeds[col] (get_global_id(0) inside a 2-D
 * launch), i.e. row r of column c consumes draws 2r+1, 2r+2 of that stream. */
void oracle_a10_initTrace(int* seeds, void* rays_, void* pois_, const float* bound8, const float* cam16,
                          float focal_length, float lens_rad, unsigned rpp, size_t gx, size_t gy) {
    cam_t cam = ld_cam(cam16);
    box_t bound = ld_box(bound8);
    pto_ray* rays = (pto_ray*)rays_;
    pto_poi* pois = (pto_poi*)pois_;
    if (rpp > 1) {
        /* no RNG draws on this branch: pixels are independent, rows go to threads */
        const uint32_t side = cln_f2u(cln_sqrt((float)rpp));
        const float delta = 1.0f / (float)side;
        const long rows = (long)(gy < cam.rows ? gy : cam.rows);
        const size_t cols = gx < cam.cols ? gx : cam.cols;
        FL(2);
#if defined(_OPENMP) && !defined(PTO_COUNT_FLOPS)
#pragma omp parallel for schedule(static, 4)
#endif
        for (long row = 0; row < rows; ++row) {
            for (size_t col = 0; col < cols; ++col) {
                size_t base = ((size_t)cam.cols * (size_t)row + col) * rpp;
                v3 fp = focal_point(&cam, (float)(uint32_t)col, (float)(uint32_t)row, focal_length);
                FL(1);
                float cy = delta / 2.0f;
                for (uint32_t i = 0; i < side; ++i) {
                    FL(1);
                    float cx = delta / 2.0f;
                    for (uint32_t j = 0; j < side; ++j) {
                        clip_and_store(&rays[base + (size_t)i * side + j], thin_lens_ray(&cam, fp, lens_rad, cx, cy), &bound);
                        FL(1);
                        cx += delta;
                    }
                    FL(1);
                    cy += delta;
                }
                for (unsigned i = 0; i < rpp; ++i) {
                    pois[base + i].matId = -1;
                    pois[base + i].atte[0] = 1.0f; pois[base + i].atte[1] = 1.0f; pois[base + i].atte[2] = 1.0f;
                }
            }
        }
        return;
    }
    /* rpp == 1: strictly sequential, row-major -- the order defines which draws a pixel gets */
    for (size_t row = 0; row < gy; ++row) {
        for (size_t col = 0; col < gx; ++col) {
            if (col >= cam.cols || row >= cam.rows) continue;
            size_t base = (size_t)cam.cols * row + col;
            v3 fp = focal_point(&cam, (float)(uint32_t)col, (float)(uint32_t)row, focal_length);
            float cy = get_rand(&seeds[col]);
            float cx = get_rand(&seeds[col]);
            clip_and_store(&rays[base], thin_lens_ray(&cam, fp, lens_rad, cx, cy), &bound);
            pois[base].matId = -1;
            pois[base].atte[0] = 1.0f; pois[base].atte[1] = 1.0f; pois[base].atte[2] = 1.0f;
        }
    }
}

/* code.cl:545-598 */
void oracle_a10_bouncePaths(void* pois_, void* rays_, int* seeds, unsigned total, size_t gsz) {
    const pto_poi* pois = (const pto_poi*)pois_;
    pto_ray* rays = (pto_ray*)rays_;
    size_t n = gsz < total ? gsz : total;
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        const pto_poi* poi = &pois[id];
        if (poi->matId >= 0) {
            v3 nrm = ld3(poi->n);
            v3 N = V(cln_fabs(nrm.x), cln_fabs(nrm.y), cln_fabs(nrm.z));
            v3 B = nrm;
            float nmin = cln_min(cln_min(N.x, N.y), N.z);
            if (N.x == nmin) B.x = 1.0f;
            else if (N.y == nmin) B.y = 1.0f;
            else B.z = 1.0f;
            N = nrm;
            B = norm3(B);
            v3 T = cross3(B, N);
            B = cross3(N, T);
            float sx = get_rand(&seeds[id]);
            float sy = get_rand(&seeds[id]);
            concentric(sx, sy, &sx, &sy);
            FL(5);
            float sz = cln_sqrt(cln_max(0.0f, cln_fma(-sy, sy, cln_fma(-sx, sx, 1.0f))));   /* code.cl:568 */
            ray_t r;
            r.o = ld3(poi->p);
            r.d = norm3(fma3(sz, N, fma3(sx, T, scl(sy, B))));                               /* code.cl:574 */
            r.mint = 0.0f;
            r.maxt = PT_INF;
            st_ray(&rays[id], r);
        } else {
            /* the reference stores an otherwise uninitialised Ray here; only mint/maxt are defined */
            rays[id].mint = PT_INF;
            rays[id].maxt = PT_INF;
        }
    }
}

/* code.cl:391-403, 600-629 */
void oracle_a10_lightRender(void* pois_, void* rays_, void* acu_, const float* light, unsigned total, size_t gsz) {
    pto_poi* pois = (pto_poi*)pois_;
    pto_ray* rays = (pto_ray*)rays_;
    float* acu = (float*)acu_;
    size_t n = gsz < total ? gsz : total;
    v3 lpos = ld3(light), lnor = ld3(light + 3);
    float radius = light[9];
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        ray_t ray = ld_ray(&rays[id]);
        if (ray.mint == ray.maxt) continue;
        v3 irr = norm3(ld3(light + 6));
        float den = dot3(ray.d, lnor);
        if (den == 0.0f) continue;
        float num = dot3(sub(lpos, ray.o), lnor);
        if (num == 0.0f) continue;
        FL(1);
        float t = num / den;
        v3 p = get_point(ray.o, t, ray.d);
        if (len3(sub(p, lpos)) > radius) continue;
        if (t >= ray.maxt) continue;
        rays[id].mint = PT_INF;
        rays[id].maxt = PT_INF;
        pois[id].matId = -1;
        FL(4);
        acu[4 * id + 0] += irr.x; acu[4 * id + 1] += irr.y; acu[4 * id + 2] += irr.z; acu[4 * id + 3] += 1.0f;
    }
}

/* code.cl:121-129, 631-673 */
void oracle_a10_initShadowTrace(void* shadow_, void* pois_, unsigned total, const float* light, int* seeds, size_t gsz) {
    pto_ray* shadow = (pto_ray*)shadow_;
    const pto_poi* pois = (const pto_poi*)pois_;
    size_t n = gsz < total ? gsz : total;
    v3 lpos0 = ld3(light), T = ld3(light + 3), B = ld3(light + 6);
    float radius = light[9];
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        const pto_poi* poi = &pois[id];
        if (poi->matId < 0) {
            shadow[id].mint = PT_INF;
            shadow[id].maxt = PT_INF;
            continue;
        }
        v3 p = fma3(0.001f, ld3(poi->n), ld3(poi->p));                /* code.cl:659: p += normal * 0.001f */
        float x = get_rand(&seeds[id]);
        float y = get_rand(&seeds[id]);
        concentric(x, y, &x, &y);
        FL(2);
        x = x * radius;
        y = y * radius;
        v3 lpos = add(lpos0, fma3(x, T, scl(y, B)));                 /* code.cl:666: light_pos += x*T + y*B */
        ray_t r;
        v3 to = sub(lpos, p);
        r.o = p;
        r.d = norm3(to);
        r.mint = 0.0f;
        r.maxt = len3(sub(lpos, p));
        st_ray(&shadow[id], r);
    }
}

static void closest_kernel(unsigned total, pto_poi* pois, pto_ray* rays, const float* prims, const float* normals,
                           const unsigned* matid, unsigned mesh_matid, const unsigned* cell_off, const float* bound8,
                           unsigned n_slabs, size_t gsz, int kind) {
    box_t bound = ld_box(bound8);
    size_t n = gsz < total ? gsz : total;
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        ray_t ray = ld_ray(&rays[id]);
        if (ray.mint == ray.maxt) continue;
        boxhit_t bh = inter_aabb(&ray, &bound);
        if (!bh.v) continue;
        champ_t ch = grid_trace(ray, bh, &bound, n_slabs, cell_off, prims, kind, 0);
        if (ch.idx == UINT_MAX) continue;
        rays[id].maxt = ch.t;
        v3 p = get_point(ray.o, ch.t, ray.d);
        v3 nrm;
        if (kind == PRIM_SPHERE) {
            nrm = norm3(sub(p, ld3(prims + 4u * (size_t)ch.idx)));       /* code.cl:794-797 */
        } else {
            const float* nn = normals + 12u * (size_t)ch.idx;             /* code.cl:405-411, 927-931 */
            FL(2);
            float w = 1.0f - ch.beta - ch.gamma;
            nrm = norm3(fma3(ch.gamma, ld3(nn + 8), fma3(w, ld3(nn), scl(ch.beta, ld3(nn + 4)))));   /* code.cl:410 */
        }
        /* trace kernels write p, normal, matId -- never atte (SURVEY 8a hazard 2) */
        st3(pois[id].p, p);
        st3(pois[id].n, nrm);
        pois[id].matId = (int32_t)(matid ? matid[ch.idx] : mesh_matid);
    }
}

/* code.cl:675-800 */
void oracle_a10_sphereTrace(unsigned total, void* pois, void* rays, void* spheres, unsigned* matid,
                            unsigned* box, const float* bound, unsigned n, size_t gsz) {
    closest_kernel(total, (pto_poi*)pois, (pto_ray*)rays, (const float*)spheres, NULL, matid, 0, box, bound, n, gsz, PRIM_SPHERE);
}
/* code.cl:802-935 */
void oracle_a10_triangleTrace(unsigned total, void* pois, void* rays, void* pos, void* nor, unsigned* matid,
                              unsigned* box, const float* bound, unsigned n, size_t gsz) {
    closest_kernel(total, (pto_poi*)pois, (pto_ray*)rays, (const float*)pos, (const float*)nor, matid, 0, box, bound, n, gsz, PRIM_TRIANGLE);
}
/* code.cl:937-1070 */
void oracle_a10_meshTrace(unsigned total, void* pois, void* rays, void* pos, void* nor, unsigned* box,
                          unsigned matid, const float* bound, unsigned n, size_t gsz) {
    closest_kernel(total, (pto_poi*)pois, (pto_ray*)rays, (const float*)pos, (const float*)nor, NULL, matid, box, bound, n, gsz, PRIM_TRIANGLE);
}

static void anyhit_kernel(unsigned total, pto_ray* shadow, const float* prims, const unsigned* cell_off,
                          const float* bound8, unsigned n_slabs, size_t gsz, int kind) {
    box_t bound = ld_box(bound8);
    size_t n = gsz < total ? gsz : total;
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        ray_t ray = ld_ray(&shadow[id]);
        if (ray.mint == ray.maxt) continue;
        boxhit_t bh = inter_aabb(&ray, &bound);
        if (!bh.v) continue;
        champ_t ch = grid_trace(ray, bh, &bound, n_slabs, cell_off, prims, kind, 1);
        shadow[id].maxt = ch.t;                       /* free way: unchanged value re-stored (code.cl:1189-1192) */
        if (ch.idx != UINT_MAX) shadow[id].mint = ch.t; /* blocked: mint == maxt marks the ray dead */
    }
}

/* code.cl:1073-1193 */
void oracle_a10_sphereShadowTrace(unsigned total, void* shadow, void* spheres, unsigned* box,
                                  const float* bound, unsigned n, size_t gsz) {
    anyhit_kernel(total, (pto_ray*)shadow, (const float*)spheres, box, bound, n, gsz, PRIM_SPHERE);
}
/* code.cl:1195-1321 */
void oracle_a10_triangleShadowTrace(unsigned total, void* shadow, void* pos, unsigned* box,
                                    const float* bound, unsigned n, size_t gsz) {
    anyhit_kernel(total, (pto_ray*)shadow, (const float*)pos, box, bound, n, gsz, PRIM_TRIANGLE);
}

/* code.cl:1323-1364 */
void oracle_a10_sceneRender(void* acu_, void* pois_, void* shadow_, void* material_, const float* light,
                            unsigned total, size_t gsz) {
    float* acu = (float*)acu_;
    pto_poi* pois = (pto_poi*)pois_;
    const pto_ray* shadow = (const pto_ray*)shadow_;
    const float* material = (const float*)material_;
    size_t n = gsz < total ? gsz : total;
    v3 lpos = ld3(light), lnor = ld3(light + 3), es = ld3(light + 6);
    float area = light[9];
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        pto_poi* poi = &pois[id];
        if (poi->matId < 0) continue;
        v3 shade = V(0.0f, 0.0f, 0.0f);
        if (shadow[id].maxt != shadow[id].mint) {
            v3 sd = ld3(shadow[id].d);
            float r = len3(sub(ld3(poi->p), lpos));
            float cosx = cln_clamp(dot3(sd, ld3(poi->n)), 0.0f, 1.0f);
            float cosy = cln_clamp(dot3(V(-sd.x, -sd.y, -sd.z), lnor), 0.0f, 1.0f);
            FL(4);
            shade = scl(area * ((cosx * cosy) / (r * r)), es);
        }
        v3 color = ld3(material + 4u * (size_t)poi->matId);
        v3 atte = ld3(poi->atte);
        st3(poi->atte, mulv(atte, color));
        v3 c = mulv(mulv(color, atte), shade);
        FL(4);
        acu[4 * id + 0] += c.x; acu[4 * id + 1] += c.y; acu[4 * id + 2] += c.z; acu[4 * id + 3] += 1.0f;
    }
}

/* code.cl:1366-1386 */
void oracle_a10_copyToPixel(void* pixel_, void* acu_, float m, unsigned pixels, unsigned rpp, size_t gsz) {
    uint8_t* pixel = (uint8_t*)pixel_;
    const float* acu = (const float*)acu_;
    size_t n = gsz < pixels ? gsz : pixels;
    PAR_FOR
    for (size_t id = 0; id < n; ++id) {
        const float* a = acu + 4u * id * (size_t)rpp;
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (unsigned i = 0; i < rpp; ++i)
            for (int k = 0; k < 4; ++k) { FL(1); c[k] = c[k] + a[4u * i + k]; }
        FL(1);
        float s = 255.0f * m;
        for (int k = 0; k < 4; ++k) {
            FL(2);
            c[k] = c[k] * s;
            c[k] = c[k] * 1.8f;
            c[k] = cln_clamp(c[k], 0.0f, 255.0f);
        }
        pixel[4 * id + 0] = (uint8_t)cln_f2u(c[0]);
        pixel[4 * id + 1] = (uint8_t)cln_f2u(c[1]);
        pixel[4 * id + 2] = (uint8_t)cln_f2u(c[2]);
        pixel[4 * id + 3] = 255;
    }
}

float oracle_bi_sin(float x) { return cln_sin(x); }
float oracle_bi_cos(float x) { return cln_cos(x); }
void oracle_set_hw_tables(const signed char* rsq_delta, const signed char* sqrt_delta) { cln_hw_tables((const int8_t*)rsq_delta, (const int8_t*)sqrt_delta); }

/* element-wise probes of the CPU model, for tests/test_ref_gpu.py (op names as in oracle/probe/builtins.cl) */
void oracle_bi_eval(const char* op, const float* a, const float* b, const float* c, float* o, size_t n) {
#define EACH for (size_t i = 0; i < n; ++i)
    if (!strcmp(op, "sqrt")) EACH o[i] = cln_sqrt(a[i]);
    else if (!strcmp(op, "sin")) EACH o[i] = cln_sin(a[i]);
    else if (!strcmp(op, "cos")) EACH o[i] = cln_cos(a[i]);
    else if (!strcmp(op, "fabs")) EACH o[i] = cln_fabs(a[i]);
    else if (!strcmp(op, "f2i")) EACH { int32_t v = cln_f2i(a[i]); memcpy(&o[i], &v, 4); }
    else if (!strcmp(op, "f2u")) EACH { uint32_t v = cln_f2u(a[i]); memcpy(&o[i], &v, 4); }
    else if (!strcmp(op, "div")) EACH o[i] = a[i] / b[i];
    else if (!strcmp(op, "fmin")) EACH o[i] = cln_fmin(a[i], b[i]);
    else if (!strcmp(op, "fmax")) EACH o[i] = cln_fmax(a[i], b[i]);
    else if (!strcmp(op, "min")) EACH o[i] = cln_min(a[i], b[i]);
    else if (!strcmp(op, "max")) EACH o[i] = cln_max(a[i], b[i]);
    else if (!strcmp(op, "mad")) EACH o[i] = cln_mad(a[i], b[i], c[i]);
    else if (!strcmp(op, "clamp")) EACH o[i] = cln_clamp(a[i], b[i], c[i]);
    else if (!strcmp(op, "muladd")) EACH {
        o[4 * i] = cln_fma(a[i], b[i], c[i]); o[4 * i + 1] = cln_fma(a[i], b[i], -c[i]);
        o[4 * i + 2] = cln_fma(-a[i], b[i], c[i]); o[4 * i + 3] = cln_fma(a[i], b[i], c[i] * a[i]);
    }
    else if (!strcmp(op, "dot")) EACH o[i] = cln_dot3(a[3 * i], a[3 * i + 1], a[3 * i + 2], b[3 * i], b[3 * i + 1], b[3 * i + 2]);
    else if (!strcmp(op, "cross")) EACH cln_cross3(a + 3 * i, b + 3 * i, o + 3 * i);
    else if (!strcmp(op, "length")) EACH o[i] = cln_length3(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
    else if (!strcmp(op, "distance")) EACH o[i] = cln_length3(a[3 * i] - b[3 * i], a[3 * i + 1] - b[3 * i + 1], a[3 * i + 2] - b[3 * i + 2]);
    else if (!strcmp(op, "normalize")) EACH cln_normalize3(a + 3 * i, o + 3 * i);
    else { fprintf(stderr, "oracle_bi_eval: unknown op %s\n", op); abort(); }
#undef EACH
}

/* ========================================================================== *
 *  Single-frame kernels of the earlier assignments (BASELINE configs 1-3).   *
 *  "A01/A04/A07 code.cl:NNN" = AssignNN-*\/code.cl.                          *
 * ========================================================================== */

/* (uchar) of a float as the compiled reference does it: truncate to int32, keep the low byte */
static inline uint8_t f2u8(float f) { return (uint8_t)(cln_f2i(f) & 0xFF); }

/* A01 code.cl:116-147 with getRay :48-59 and interSphere :61-91.  The camera carries rows, cols as FLOATS
 * in .sE, .sF (swapped w.r.t. A02+).  The reference has no range check; it is launched on a padded NDRange
 * and relies on width/height being multiples of the work-group shape.  We skip out-of-image work-items. */
void oracle_a01_raytrace(void* pixels_, const float* cam, size_t gx, size_t gy) {
    uint8_t* pixels = (uint8_t*)pixels_;
    const v3 eye = ld3(cam), U = ld3(cam + 3), Vv = ld3(cam + 6), W = ld3(cam + 9);
    const float width = cam[12], height = cam[13], rows = cam[14], cols = cam[15];
    const uint32_t ucols = cln_f2u(cols);
    for (size_t row = 0; row < gy; ++row)
        for (size_t col = 0; col < gx; ++col) {
            if (!((float)col < cols) || !((float)row < rows)) continue;
            float sx = (-0.5f + ((float)(uint32_t)col + 0.5f) / cols) * width;
            float sy = (0.5f - ((float)(uint32_t)row + 0.5f) / rows) * height;
            v3 cop = add(fma3(sx, U, scl(sy, Vv)), scl(-1.0f, W));   /* A01 code.cl:54-56 */
            v3 o = eye;
            v3 d = norm3(sub(cop, o));
            const v3 sc = V(0.0f, 0.0f, 1.0f);
            const float sr = 0.5f;
            v3 omc = sub(o, sc);
            float a = dot3(d, d);
            float b = 2.0f * dot3(omc, d);
            float c = cln_fma(-sr, sr, dot3(omc, omc));             /* A01 code.cl:68 */
            float dis = cln_fma(b, b, -((4.0f * a) * c));           /* A01 code.cl:69 */
            int v = 0;
            float t = PT_INF;
            if (!(dis < 0.0f)) {
                float sq = cln_sqrt(dis);
                float t0 = (-b - sq) / 2 * a;   /* sic: (x / 2) * a */
                float t1 = (-b + sq) / 2 * a;
                if (t0 > 0.0f && t0 < PT_INF) { t = t0; v = 1; }
                else if (t1 > 0.0f && t1 < PT_INF) { t = t1; v = 1; }
            }
            uint8_t base = v ? f2u8((1.0f - t) * 255.0f) : 0;
            uint8_t* px = pixels + 4u * ((size_t)ucols * row + col);
            px[0] = base; px[1] = base; px[2] = base; px[3] = 255;
        }
}

/* A04 code.cl:86-97 == A07 code.cl:97-108: pinhole ray through the pixel centre */
static ray_t pinhole_ray(const cam_t* c, float col, float row) { return get_ray(c, col, row); }

/* A04 code.cl:204-215 (clip = 0) and A07 code.cl:311-335 (clip = 1) */
static void frame_init(uint8_t* pixels, const float* cam16, pto_ray* rays, const float* bound8, size_t gx, size_t gy, int clip) {
    cam_t cam = ld_cam(cam16);
    box_t bound;
    if (clip) bound = ld_box(bound8);
    for (size_t row = 0; row < gy; ++row)
        for (size_t col = 0; col < gx; ++col) {
            if (col >= cam.cols || row >= cam.rows) continue;
            ray_t r = pinhole_ray(&cam, (float)(uint32_t)col, (float)(uint32_t)row);
            size_t pix = (size_t)cam.cols * row + col;
            if (clip) clip_and_store(&rays[pix], r, &bound); else st_ray(&rays[pix], r);
            pixels[4 * pix + 0] = 0; pixels[4 * pix + 1] = 0; pixels[4 * pix + 2] = 0; pixels[4 * pix + 3] = 255;
        }
}
void oracle_a04_initTrace(void* pixels, const float* cam, void* rays, size_t gx, size_t gy) {
    frame_init((uint8_t*)pixels, cam, (pto_ray*)rays, NULL, gx, gy, 0);
}
void oracle_a07_initTrace(void* pixels, const float* cam, void* rays, const float* bound, size_t gx, size_t gy) {
    frame_init((uint8_t*)pixels, cam, (pto_ray*)rays, bound, gx, gy, 1);
}

/* interTriangle of A04 (code.cl:146-184: gamma > 1 also rejects) and A07 (code.cl:165-203); both accept t in the
 * OPEN interval (mint, maxt), unlike A10 */
static int inter_triangle_open(const ray_t* r, const float* tp, int gamma_le_1, float* t_out, float* beta_out, float* gamma_out) {
    v3 p0 = ld3(tp), p1 = ld3(tp + 4), p2 = ld3(tp + 8);
    v3 e1 = sub(p1, p0);
    v3 e2 = sub(p2, p0);
    float div = dot3(cross3(e2, e1), r->d);
    if (div <= 0) return 0;
    float idiv = 1.0f / div;
    v3 s = sub(r->o, p0);
    float beta = dot3(cross3(s, r->d), e2) * idiv;
    if (beta < 0.0f || beta > 1.0f) return 0;
    float gamma = dot3(cross3(s, e1), r->d) * idiv;
    float gb = gamma + beta;
    if (gamma < 0.0f || (gamma_le_1 && gamma > 1.0f) || gb < 0.0f || gb > 1.0f) return 0;
    float t = dot3(cross3(s, e2), e1) * -idiv;
    if (t > r->mint && t < r->maxt) { *t_out = t; *beta_out = beta; *gamma_out = gamma; return 1; }
    return 0;
}
static v3 interp_normal(const float* normals, uint32_t i, float beta, float gamma) {
    const float* nn = normals + 12u * (size_t)i;
    float w = 1.0f - beta - gamma;
    return norm3(fma3(gamma, ld3(nn + 8), fma3(w, ld3(nn), scl(beta, ld3(nn + 4)))));   /* A04 code.cl:193, A07 code.cl:300 */
}

/* A04 code.cl:262-315: every pixel against every triangle */
void oracle_a04_meshTrace(void* pixels_, const float* cam16, void* rays_, unsigned t_size, void* pos_, void* nor_, unsigned* mindex,
                          void* mcolor_, size_t gx, size_t gy) {
    uint8_t* pixels = (uint8_t*)pixels_;
    pto_ray* rays = (pto_ray*)rays_;
    const float *pos = (const float*)pos_, *nor = (const float*)nor_, *mcolor = (const float*)mcolor_;
    cam_t cam = ld_cam(cam16);
    const long rows = (long)(gy < cam.rows ? gy : cam.rows);
    const size_t cols = gx < cam.cols ? gx : cam.cols;
    PAR_ROWS
    for (long row = 0; row < rows; ++row)
        for (size_t col = 0; col < cols; ++col) {
            size_t pix = (size_t)cam.cols * (size_t)row + col;
            ray_t ray = ld_ray(&rays[pix]);
            float champ_t = PT_INF, cb = 0.0f, cg = 0.0f;
            unsigned champ_i = t_size;
            for (unsigned i = 0; i < t_size; ++i) {
                float t, b, g;
                if (inter_triangle_open(&ray, pos + 12u * (size_t)i, 1, &t, &b, &g) && t < champ_t) { champ_t = t; champ_i = i; cb = b; cg = g; }
            }
            if (champ_i >= t_size) continue;
            rays[pix].maxt = champ_t;
            v3 n = interp_normal(nor, champ_i, cb, cg);
            float shade = cln_clamp(dot3(cam.W, n), 0.0f, 1.0f);
            const float* mc = mcolor + 4u * (size_t)mindex[champ_i];
            pixels[4 * pix + 0] = f2u8((mc[0] * 255.0f) * shade);
            pixels[4 * pix + 1] = f2u8((mc[1] * 255.0f) * shade);
            pixels[4 * pix + 2] = f2u8((mc[2] * 255.0f) * shade);
            pixels[4 * pix + 3] = 255;
        }
}

/* A07 code.cl:475-626: 3-D grid DDA (same walk as A10), colour = parity of the hit cell x fake shade */
void oracle_a07_meshTrace(void* pixels_, const float* cam16, void* rays_, unsigned t_size, void* pos_, void* nor_, unsigned* mindex,
                          void* mcolor, const float* bound8, unsigned n_slabs, unsigned* slab_size, size_t gx, size_t gy) {
    (void)t_size; (void)mindex; (void)mcolor;   /* bound by the host, unused by the kernel's live code */
    uint8_t* pixels = (uint8_t*)pixels_;
    pto_ray* rays = (pto_ray*)rays_;
    const float *pos = (const float*)pos_, *nor = (const float*)nor_;
    cam_t cam = ld_cam(cam16);
    box_t bound = ld_box(bound8);
    const long rows = (long)(gy < cam.rows ? gy : cam.rows);
    const size_t cols = gx < cam.cols ? gx : cam.cols;
    const uint32_t n = n_slabs, zs = n * n, ys = n;
    PAR_ROWS
    for (long row = 0; row < rows; ++row)
        for (size_t col = 0; col < cols; ++col) {
            size_t pix = (size_t)cam.cols * (size_t)row + col;
            ray_t ray = ld_ray(&rays[pix]);
            if (ray.mint == ray.maxt) continue;
            boxhit_t bh = inter_aabb(&ray, &bound);
            if (!bh.v) continue;
            axis_t ax = axis_setup(ray.o.x, ray.d.x, bh.tmin, bound.pmin.x, bound.pmax.x, n);
            axis_t ay = axis_setup(ray.o.y, ray.d.y, bh.tmin, bound.pmin.y, bound.pmax.y, n);
            axis_t az = axis_setup(ray.o.z, ray.d.z, bh.tmin, bound.pmin.z, bound.pmax.z, n);
            float champ_t = ray.maxt, cb = 0.0f, cg = 0.0f, t = bh.tmin;
            uint32_t champ_i = UINT_MAX;
            int hx = 0, hy = 0, hz = 0;
            for (;;) {
                ray.mint = t;
                ray.maxt = cln_min(cln_min(ax.tnext, ay.tnext), az.tnext);
                uint32_t cell = (uint32_t)az.slab * zs + (uint32_t)ay.slab * ys + (uint32_t)ax.slab;
                for (uint32_t i = slab_size[cell]; i < slab_size[cell + 1]; ++i) {
                    float ti, b, g;
                    if (inter_triangle_open(&ray, pos + 12u * (size_t)i, 0, &ti, &b, &g) && ti < champ_t) {
                        champ_t = ti; champ_i = i; cb = b; cg = g; hx = ax.slab; hy = ay.slab; hz = az.slab;
                    }
                }
                if (champ_i != UINT_MAX) break;
                t = ray.maxt;
                if (t == ax.tnext) { ax.tnext += ax.dt; if (t >= bh.tmax) break; ax.slab += ax.dslab; if (ax.slab == ax.limit) break; }
                else if (t == ay.tnext) { ay.tnext += ay.dt; if (t >= bh.tmax) break; ay.slab += ay.dslab; if (ay.slab == ay.limit) break; }
                else { az.tnext += az.dt; if (t >= bh.tmax) break; az.slab += az.dslab; if (az.slab == az.limit) break; }
            }
            if (champ_i == UINT_MAX) continue;
            rays[pix].maxt = champ_t;
            v3 nrm = interp_normal(nor, champ_i, cb, cg);
            float shade = cln_clamp(dot3(cam.W, nrm), 0.0f, 1.0f);
            float k = shade * 127.0f;
            pixels[4 * pix + 0] = f2u8((float)((hx % 2) + 1) * k);
            pixels[4 * pix + 1] = f2u8((float)((hy % 2) + 1) * k);
            pixels[4 * pix + 2] = f2u8((float)((hz % 2) + 1) * k);
            pixels[4 * pix + 3] = 255;
        }
}

/* A07 code.cl:337-473: the grid walk over atoms {c, r*r}; the champion keeps its CELL (champ_slab, :402, :425-429), whose
 * parity colours the pixel; shade = clamp(dot(W, normalize(p - c))) (:455-469).  s_mindex / m_color are bound, never read. */
void oracle_a07_molTrace(void* pixels_, const float* cam16, void* rays_, unsigned s_size, void* atoms_, unsigned* mindex, void* mcolor,
                         const float* bound8, unsigned n_slabs, unsigned* slab_size, size_t gx, size_t gy) {
    (void)s_size; (void)mindex; (void)mcolor;
    uint8_t* pixels = (uint8_t*)pixels_;
    pto_ray* rays = (pto_ray*)rays_;
    const float* atoms = (const float*)atoms_;
    cam_t cam = ld_cam(cam16);
    box_t bound = ld_box(bound8);
    const long rows = (long)(gy < cam.rows ? gy : cam.rows);
    const size_t cols = gx < cam.cols ? gx : cam.cols;
    const uint32_t n = n_slabs, zs = n * n, ys = n;
    PAR_ROWS
    for (long row = 0; row < rows; ++row)
        for (size_t col = 0; col < cols; ++col) {
            size_t pix = (size_t)cam.cols * (size_t)row + col;
            ray_t ray = ld_ray(&rays[pix]);
            if (ray.mint == ray.maxt) continue;
            boxhit_t bh = inter_aabb(&ray, &bound);
            if (!bh.v) continue;
            axis_t ax = axis_setup(ray.o.x, ray.d.x, bh.tmin, bound.pmin.x, bound.pmax.x, n);
            axis_t ay = axis_setup(ray.o.y, ray.d.y, bh.tmin, bound.pmin.y, bound.pmax.y, n);
            axis_t az = axis_setup(ray.o.z, ray.d.z, bh.tmin, bound.pmin.z, bound.pmax.z, n);
            float champ_t = ray.maxt, t = bh.tmin;
            uint32_t champ_i = UINT_MAX;
            int hx = 0, hy = 0, hz = 0;
            for (;;) {
                ray.mint = t;
                ray.maxt = cln_min(cln_min(ax.tnext, ay.tnext), az.tnext);
                uint32_t cell = (uint32_t)az.slab * zs + (uint32_t)ay.slab * ys + (uint32_t)ax.slab;
                for (uint32_t i = slab_size[cell]; i < slab_size[cell + 1]; ++i) {
                    float ti;
                    if (inter_sphere(&ray, atoms + 4u * (size_t)i, &ti) && ti < champ_t) {
                        champ_t = ti; champ_i = i; hx = ax.slab; hy = ay.slab; hz = az.slab;
                    }
                }
                if (champ_i != UINT_MAX) break;
                t = ray.maxt;
                if (t == ax.tnext) { ax.tnext += ax.dt; if (t >= bh.tmax) break; ax.slab += ax.dslab; if (ax.slab == ax.limit) break; }
                else if (t == ay.tnext) { ay.tnext += ay.dt; if (t >= bh.tmax) break; ay.slab += ay.dslab; if (ay.slab == ay.limit) break; }
                else { az.tnext += az.dt; if (t >= bh.tmax) break; az.slab += az.dslab; if (az.slab == az.limit) break; }
            }
            if (champ_i == UINT_MAX) continue;
            rays[pix].maxt = champ_t;
            const float* a = atoms + 4u * (size_t)champ_i;
            v3 ip = get_point(ray.o, champ_t, ray.d);
            v3 c = {a[0], a[1], a[2]};
            float shade = cln_clamp(dot3(cam.W, norm3(sub(ip, c))), 0.0f, 1.0f);
            float k = shade * 127.0f;
            pixels[4 * pix + 0] = f2u8((float)((hx % 2) + 1) * k);
            pixels[4 * pix + 1] = f2u8((float)((hy % 2) + 1) * k);
            pixels[4 * pix + 2] = f2u8((float)((hz % 2) + 1) * k);
            pixels[4 * pix + 3] = 255;
        }
}
